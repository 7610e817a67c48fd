// bgnorm.hip — prl::backgroundNormalization for pages resident in device memory.
//
// Reference: src/backgroundNormalization.cpp:36-61 = opencvToLeptonica -> pixBackgroundNormSimple(pixs, NULL, NULL) ->
// leptonicaToOpenCV, with the channel conventions of src/formatConvert.cpp:38-218 (1 channel -> 8 bpp; 3 / 4 channels ->
// 32 bpp with Mat byte i in slot i, three channels out).  The arithmetic is Leptonica's adaptmap.c / convolve.c
// [upstream, restated function by function in the test oracle (oracle/, bgnorm file)]: tile 10 x 15, foreground
// threshold 60, mincount 40, bgval 200, smoothing 2 x 1.
//
//   k_bg_tiles<CH>  foreground mask (pixel < 60 on the gray / green channel, dilated 7 x 7) and the per-tile average of
//                   the pixels outside it, complete tiles only.  One workgroup = one tile row x 25 tiles: 21 rows x 256
//                   columns of threshold flags staged in LDS (the 3-pixel halo of the dilation), one thread per page
//                   column, the 10 columns of a tile reduced through LDS.  HBM: the gray channel 1.4x (halo rows) + every
//                   channel once.
//   k_bg_maps       per (page, channel), one workgroup: pixFillMapHoles (column fill, column replication, last column),
//                   pixBlockconv(2, 1) with blockconvLow's edge rule and float32 rescaling, 16-bit inverse map.  The maps
//                   are tiny (248 x 234 for A4) and live in L2.
//   k_bg_apply<CH>  out = min(255, p * inv[y/15][x/10] >> 8), 16 pixels per thread; a page whose map could not be made is
//                   copied (Leptonica returns a copy of the source).  1 B read + 1 B written per channel byte.
// Integer throughout except blockconvLow's float32 factors (one rounding per operation, -ffp-contract=off): bit-exact
// against the oracle.
#include <algorithm>

#include "prl_internal.h"

namespace prl_hip {
namespace {

constexpr int SX = 10, SY = 15, THRESH = 60, MINCOUNT = 40, BGVAL = 200, WC = 2, HC = 1;
constexpr int TILES_PER_BLOCK = 25;                 // 250 page columns + 6 halo columns = 256 threads
constexpr int COLS_PER_BLOCK = TILES_PER_BLOCK * SX;
constexpr int FLAG_ROWS = SY + 6;

struct BgGeom {
    int width, height, mw, mh, nx, ny;
    size_t map_page;   // bytes of one page's u8 maps  (och * mw * mh, rounded up)
    size_t inv_page;   // elements of one page's u16 inverse maps
};

// One workgroup = one tile row (15 page rows) x 25 tiles = 250 page columns + the 3-column halo of the dilation on both
// sides: thread t owns page column x0 - 3 + t, reads its 21 rows (15 + the 3-row halo above and below) ONCE, keeps the
// threshold flags of those rows as a 21-bit word and the 15 rows' pixel values in registers.  The 7 x 1 dilation is the OR
// of seven neighbouring threads' words (LDS, one word per column, conflict-free), the 1 x 7 dilation seven shifts of the
// result - all 21 rows at once - so the foreground mask costs 7 LDS reads per column instead of the 147 byte reads of the
// first version, and the page is read once instead of 2.4 times.
template <int CH>
__global__ void __launch_bounds__(256) k_bg_tiles(PageSet src, BgGeom g, uint8_t* __restrict__ maps)
{
    constexpr int OCH = CH == 1 ? 1 : 3;
    constexpr int GCH = CH == 1 ? 0 : 1;  // pixConvertRGBToGrayFast: the green slot
    __shared__ unsigned colbits[256 + 8];
    __shared__ unsigned short colsum[OCH][COLS_PER_BLOCK];
    __shared__ uint8_t colcnt[COLS_PER_BLOCK];
    const int page = blockIdx.z, ty = blockIdx.y, t = threadIdx.x;
    const int x0 = blockIdx.x * COLS_PER_BLOCK;
    const uint8_t* base = src.page(page);
    const int x = x0 - 3 + t;
    const bool xin = x >= 0 && x < g.width;
    unsigned bits = 0;          // bit r: page row ty*15 - 3 + r of this column is below the threshold
    unsigned px[SY][OCH];       // the 15 tile rows of this column
#pragma unroll
    for (int r = 0; r < FLAG_ROWS; ++r) {
        const int y = ty * SY - 3 + r;
        unsigned v[OCH];
#pragma unroll
        for (int c = 0; c < OCH; ++c) v[c] = 255;
        if (xin && y >= 0 && y < g.height) {
            const uint8_t* p = base + (size_t)y * src.step + (size_t)x * CH;
#pragma unroll
            for (int c = 0; c < OCH; ++c) v[c] = p[c];
            bits |= (unsigned)(v[GCH] < THRESH) << r;
        }
        if (r >= 3 && r < 3 + SY) {
#pragma unroll
            for (int c = 0; c < OCH; ++c) px[r - 3][c] = v[c];
        }
    }
    colbits[t] = bits;
    if (t < 8) colbits[256 + t] = 0;
    __syncthreads();
    // thread t now plays tile column x0 + t (its own pixels belong to column x0 - 3 + t: the sums below use the owner's
    // registers, so the dilation result is computed for the OWNED column: neighbours t-3 .. t+3)
    unsigned hd = 0;
#pragma unroll
    for (int d = -3; d <= 3; ++d) {
        const int n = t + d;
        hd |= (n >= 0 && n < 256) ? colbits[n] : 0u;
    }
    // vertical 1 x 7: bit k of fg = OR of hd bits k .. k+6 (rows y-3 .. y+3 of tile row k)
    const unsigned fg = hd | (hd >> 1) | (hd >> 2) | (hd >> 3) | (hd >> 4) | (hd >> 5) | (hd >> 6);
    const int tc = t - 3;  // index of the owned column inside the block's 250 tile columns
    if (tc >= 0 && tc < COLS_PER_BLOCK) {
        unsigned sum[OCH];
#pragma unroll
        for (int c = 0; c < OCH; ++c) sum[c] = 0;
        unsigned cnt = 0;
        if (x < g.nx * SX) {
#pragma unroll
            for (int k = 0; k < SY; ++k) {
                if (((fg >> k) & 1u) == 0) {
#pragma unroll
                    for (int c = 0; c < OCH; ++c) sum[c] += px[k][c];
                    ++cnt;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < OCH; ++c) colsum[c][tc] = (unsigned short)sum[c];
        colcnt[tc] = (uint8_t)cnt;
    }
    __syncthreads();
    if (t < TILES_PER_BLOCK) {
        const int j = blockIdx.x * TILES_PER_BLOCK + t;
        if (j < g.nx) {
            unsigned cnt = 0, sum[OCH];
#pragma unroll
            for (int c = 0; c < OCH; ++c) sum[c] = 0;
#pragma unroll
            for (int m = 0; m < SX; ++m) {
                cnt += colcnt[t * SX + m];
#pragma unroll
                for (int c = 0; c < OCH; ++c) sum[c] += colsum[c][t * SX + m];
            }
            uint8_t* mp = maps + (size_t)page * g.map_page + (size_t)ty * g.mw + j;
#pragma unroll
            for (int c = 0; c < OCH; ++c) mp[(size_t)c * g.mw * g.mh] = cnt >= MINCOUNT ? (uint8_t)(sum[c] / cnt) : 0;
        }
    }
}

// One workgroup per (channel, page).  `maps` rows beyond ny / columns beyond nx are zero (hipMemsetAsync before k_bg_tiles).
__global__ void __launch_bounds__(256) k_bg_maps(BgGeom g, int och, uint8_t* __restrict__ maps, unsigned short* __restrict__ inv,
                                                 int* __restrict__ page_fail)
{
    constexpr int MAX_MW = 4096;
    __shared__ uint8_t na[MAX_MW];
    __shared__ int s_nmiss, s_goodcol;
    const int c = blockIdx.x, page = blockIdx.y, t = threadIdx.x;
    const int w = g.mw, h = g.mh, nx = g.nx, ny = g.ny;
    // the map itself is tiny (248 x 234 bytes for A4): work on an LDS copy when it fits (the column fills are chains of
    // dependent accesses: 0.61 ms for 64 gray pages' maps from memory, 0.27 ms from LDS), on the global one otherwise
    constexpr int LDS_MAP = 60000;  // an A4 map is 248 x 234 = 58 032 bytes
    __shared__ uint8_t lmap[LDS_MAP];
    uint8_t* gm = maps + (size_t)page * g.map_page + (size_t)c * w * h;
    const bool in_lds = w * h <= LDS_MAP;
    if (in_lds)
        for (int i = t; i < w * h; i += 256) lmap[i] = gm[i];
    uint8_t* m = in_lds ? lmap : gm;  // (phases are separated by workgroup barriers; no volatile: the memory path stays cached)
    if (t == 0) { s_nmiss = 0; s_goodcol = w; }
    __syncthreads();
    // pixFillMapHoles, columns
    for (int j = t; j < nx; j += 256) {
        int y = -1;
        uint8_t val = 0;
        for (int i = 0; i < ny; ++i) {
            val = m[(size_t)i * w + j];
            if (val != 0) { y = i; break; }
        }
        if (y < 0) {
            na[j] = 0;
            atomicAdd(&s_nmiss, 1);
        } else {
            na[j] = 1;
            atomicMin(&s_goodcol, j);
            for (int i = y - 1; i >= 0; --i) m[(size_t)i * w + j] = val;
            uint8_t lastval = m[j];
            for (int i = 1; i < h; ++i) {
                val = m[(size_t)i * w + j];
                if (val == 0) m[(size_t)i * w + j] = lastval;
                else lastval = val;
            }
        }
    }
    if (t == 0 && w > nx) na[nx] = 0;
    __threadfence_block();
    __syncthreads();
    const bool small_map = w < 5 || h < 5;  // pixGetInvBackgroundMap: "w and h must be >= 5"
    if (s_nmiss == nx || small_map) {       // no data in any column: the map cannot be made
        if (t == 0) atomicOr(&page_fail[page], 1);
        return;
    }
    if (s_nmiss > 0) {  // columns without data copy the nearest good column to their left (to their right before the first)
        const int goodcol = s_goodcol;
        for (int j = t; j < w; j += 256) {
            if (na[j]) continue;
            int sj = goodcol;
            if (j > goodcol) {
                sj = j - 1;
                while (!na[sj]) --sj;
            }
            for (int i = 0; i < h; ++i) m[(size_t)i * w + j] = m[(size_t)i * w + sj];
        }
        __threadfence_block();
        __syncthreads();
    }
    if (w > nx) {  // the column of incomplete tiles replicates its left neighbour
        for (int i = t; i < h; i += 256) m[(size_t)i * w + w - 1] = m[(size_t)i * w + w - 2];
        __threadfence_block();
        __syncthreads();
    }
    // pixBlockconv(2, 1) + inverse
    const int fwc = 2 * WC + 1, fhc = 2 * HC + 1, wmwc = w - WC, hmhc = h - HC;
    const float norm = (float)(1.0 / ((float)fwc * fhc));
    unsigned short* iv = inv + (size_t)page * g.inv_page + (size_t)c * w * h;
    for (int idx = t; idx < w * h; idx += 256) {
        const int i = idx / w, j = idx - i * w;
        const int imin = max(i - 1 - HC, 0), imax = min(i + HC, h - 1);
        const int jmin = max(j - 1 - WC, 0), jmax = min(j + WC, w - 1);
        unsigned sum = 0;  // a[imax][jmax] - a[imax][jmin] - a[imin][jmax] + a[imin][jmin]: rows imin+1..imax, cols jmin+1..jmax
        for (int y = imin + 1; y <= imax; ++y)
            for (int x = jmin + 1; x <= jmax; ++x) sum += m[(size_t)y * w + x];
        const float prod = norm * (float)sum;
        unsigned val = (unsigned)(uint8_t)((double)prod + 0.5);
        const bool has_h = i <= HC || i >= hmhc, has_w = j <= WC || j >= wmwc;
        if (has_h || has_w) {
            const float normh = i <= HC ? (float)fhc / (float)max(1, HC + i) : (float)fhc / (float)(HC + h - i);
            const float normw = j <= WC ? (float)fwc / (float)max(1, WC + j) : (float)fwc / (float)(WC + w - j);
            float tv = (float)val;
            if (has_h) tv = tv * normh;
            if (has_w) tv = tv * normw;
            val = (unsigned)(uint8_t)(tv < 255.0f ? tv : 255.0f);
        }
        iv[idx] = val > 0 ? (unsigned short)((256 * BGVAL) / val) : (unsigned short)(BGVAL / 2);
    }
}

// out = min(255, p * inv[y / 15][x / 10] >> 8).  One thread per group of 16 pixels, groups numbered row after row (no
// workgroup is left mostly empty at the end of every row).  A group touches at most three map columns (16 pixels, 10 per tile):
// their multipliers are fetched once and picked per pixel by two comparisons; bytes are unpacked from and packed into dwords at
// compile-time positions.  (First version: 4 pixels per thread with a map fetch and an integer division per pixel - 26 vector
// instructions per pixel, profiles/r02/pmc_stages.txt.)  A page whose map could not be made is copied: multiplier 256.
template <int CH>
struct ApplyGeo { static constexpr int APX = CH == 1 ? 16 : 8; };   // pixels per thread (3 / 4 channels: 8 measured better than 16)
template <int CH>
__global__ void __launch_bounds__(256) k_bg_apply(PageSet src, PageSetOut dst, BgGeom g, const unsigned short* __restrict__ inv,
                                                  const int* __restrict__ page_fail)
{
    constexpr int OCH = CH == 1 ? 1 : 3, APX = ApplyGeo<CH>::APX;
    const int page = blockIdx.y;
    const unsigned groups = (unsigned)(g.width + APX - 1) / (unsigned)APX;
    const unsigned gi = blockIdx.x * 256u + threadIdx.x;
    const int y = (int)(gi / groups);
    if (y >= g.height) return;
    const int x0 = (int)(gi - (unsigned)y * groups) * APX;
    const int n = min(APX, g.width - x0);
    const uint8_t* s = src.page(page) + (size_t)y * src.step + (size_t)x0 * CH;
    uint8_t* d = dst.page(page) + (size_t)y * dst.step + (size_t)x0 * OCH;
    const bool fail = page_fail[page] != 0;
    const unsigned short* iv = inv + (size_t)page * g.inv_page + (size_t)(y / SY) * g.mw;
    const size_t plane = (size_t)g.mw * g.mh;
    const int tx0 = x0 / SX, rem0 = x0 - tx0 * SX;
    unsigned m[OCH][3];
#pragma unroll
    for (int c = 0; c < OCH; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) m[c][k] = fail ? 256u : (unsigned)iv[(size_t)c * plane + min(tx0 + k, g.mw - 1)];
    if (n == APX) {
        unsigned in[APX * CH / 4], out[APX * OCH / 4];
#pragma unroll
        for (int i = 0; i < APX * CH / 4; ++i) __builtin_memcpy(&in[i], s + 4 * i, 4);   // (any alignment)
#pragma unroll
        for (int i = 0; i < APX * OCH / 4; ++i) out[i] = 0;
#pragma unroll
        for (int px = 0; px < APX; ++px) {
            const int sel = rem0 + px;   // 0 .. 24: map column tx0 + sel / 10
#pragma unroll
            for (int c = 0; c < OCH; ++c) {
                const unsigned mm = sel < SX ? m[c][0] : (sel < 2 * SX ? m[c][1] : m[c][2]);
                const int bi = px * CH + c, bo = px * OCH + c;
                const unsigned p = (in[bi / 4] >> (8 * (bi % 4))) & 0xffu;
                out[bo / 4] |= min(255u, (p * mm) >> 8) << (8 * (bo % 4));
            }
        }
#pragma unroll
        for (int i = 0; i < APX * OCH / 4; ++i) __builtin_memcpy(d + 4 * i, &out[i], 4);
    } else {
        for (int px = 0; px < n; ++px) {
            const int k = (rem0 + px) / SX;
            for (int c = 0; c < OCH; ++c) d[px * OCH + c] = (uint8_t)min(255u, ((unsigned)s[px * CH + c] * m[c][k]) >> 8);
        }
    }
}

}  // namespace

int bgnorm_run(int n_pages, int channels, const PageSet& src, int width, int height, const PageSetOut& dst, void* work,
               hipStream_t stream);
size_t bgnorm_work_bytes(int n_pages, int channels, int width, int height);

static BgGeom make_geom(int channels, int width, int height)
{
    BgGeom g{};
    const int och = channels == 1 ? 1 : 3;
    g.width = width; g.height = height;
    g.mw = (width + SX - 1) / SX; g.mh = (height + SY - 1) / SY;
    g.nx = width / SX; g.ny = height / SY;
    g.map_page = ((size_t)och * g.mw * g.mh + 255) / 256 * 256;
    g.inv_page = ((size_t)och * g.mw * g.mh + 127) / 128 * 128;
    return g;
}

size_t bgnorm_work_bytes(int n_pages, int channels, int width, int height)
{
    const BgGeom g = make_geom(channels, width, height);
    return (size_t)n_pages * (g.map_page + g.inv_page * 2) + (((size_t)n_pages * sizeof(int) + 255) / 256 * 256);
}

// work: [u8 maps][u16 inverse maps][int fail flag per page]
int bgnorm_run(int n_pages, int channels, const PageSet& src, int width, int height, const PageSetOut& dst, void* work,
               hipStream_t stream)
{
    const BgGeom g = make_geom(channels, width, height);
    const int och = channels == 1 ? 1 : 3;
    uint8_t* maps = static_cast<uint8_t*>(work);
    unsigned short* inv = reinterpret_cast<unsigned short*>(maps + (size_t)n_pages * g.map_page);
    int* fail = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(inv) + (size_t)n_pages * g.inv_page * 2);
    PRL_HIP_CHECK(hipMemsetAsync(maps, 0, (size_t)n_pages * g.map_page, stream));
    PRL_HIP_CHECK(hipMemsetAsync(fail, 0, (size_t)n_pages * sizeof(int), stream));
    if (g.nx > 0 && g.ny > 0) {
        const dim3 grid((unsigned)((g.nx + TILES_PER_BLOCK - 1) / TILES_PER_BLOCK), (unsigned)g.ny, (unsigned)n_pages);
        if (channels == 1) hipLaunchKernelGGL(k_bg_tiles<1>, grid, dim3(256), 0, stream, src, g, maps);
        else if (channels == 3) hipLaunchKernelGGL(k_bg_tiles<3>, grid, dim3(256), 0, stream, src, g, maps);
        else hipLaunchKernelGGL(k_bg_tiles<4>, grid, dim3(256), 0, stream, src, g, maps);
        PRL_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bg_maps, dim3((unsigned)och, (unsigned)n_pages), dim3(256), 0, stream, g, och, maps, inv, fail);
    PRL_HIP_CHECK(hipGetLastError());
    const int apx = channels == 1 ? ApplyGeo<1>::APX : ApplyGeo<3>::APX;
    const dim3 agrid((unsigned)(((size_t)((width + apx - 1) / apx) * height + 255) / 256), (unsigned)n_pages);
    if (channels == 1) hipLaunchKernelGGL(k_bg_apply<1>, agrid, dim3(256), 0, stream, src, dst, g, inv, fail);
    else if (channels == 3) hipLaunchKernelGGL(k_bg_apply<3>, agrid, dim3(256), 0, stream, src, dst, g, inv, fail);
    else hipLaunchKernelGGL(k_bg_apply<4>, agrid, dim3(256), 0, stream, src, dst, g, inv, fail);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

int prl_hip_bgnorm_out_channels(int channels) { return channels == 1 ? 1 : 3; }

int prl_hip_bgnorm_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;  // backgroundNormalization.cpp:40-43
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;  // formatConvert.cpp:103-104
    const int och = channels == 1 ? 1 : 3;
    if (n_pages < 0 || !d_src || !d_dst || src_step < (size_t)width * channels || dst_step < (size_t)width * och)
        return PRL_ERR_BAD_ARG;
    if (height > 65535 || width > 40950) return PRL_ERR_BAD_ARG;  // grid.y; k_bg_maps' column flags
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int chunk = std::min(n_pages, 16384);
    st = ensure_scratch(ctx, bgnorm_work_bytes(chunk, channels, width, height));
    if (st != PRL_OK) return st;
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(hs, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    for (int first = 0; first < n_pages; first += chunk) {
        PageSet s{};
        s.base = d_src + (size_t)first * src_page_stride; s.page_stride = src_page_stride; s.step = src_step;
        PageSetOut d{};
        d.base = d_dst + (size_t)first * dst_page_stride; d.page_stride = dst_page_stride; d.step = dst_step;
        st = bgnorm_run(std::min(chunk, n_pages - first), channels, s, width, height, d, ctx->scratch, hs);
        if (st != PRL_OK) break;
    }
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, hs));
    return st;
}

int prl_hip_bgnorm_host(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY;
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    const int och = channels == 1 ? 1 : 3;
    if (!dst || src_step < (size_t)width * channels || dst_step < (size_t)width * och) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    const size_t in_row = (size_t)width * channels, out_row = (size_t)width * och;
    const size_t in_bytes = (in_row * (size_t)height + 255) / 256 * 256, out_bytes = out_row * (size_t)height;
    std::lock_guard<std::mutex> slk(ctx->stage_mu);
    st = ensure_stage(ctx, in_bytes + out_bytes);
    if (st != PRL_OK) return st;
    st = ensure_stage_pinned(ctx, in_bytes + out_bytes);
    if (st != PRL_OK) return st;
    uint8_t* d_in = static_cast<uint8_t*>(ctx->stage);
    uint8_t* d_out = d_in + in_bytes;
    hipStream_t stream = nullptr;
    DrainOnExit drain_guard{stream};   // (direct DMA from the caller's pinned page: see prl_internal.h)
    st = stage_upload(ctx, 0, src, src_step, in_row, height, d_in, stream);
    if (st != PRL_OK) return st;
    st = prl_hip_bgnorm_batch_device(1, channels, d_in, in_bytes, in_row, width, height, d_out, out_bytes, out_row, stream);
    if (st != PRL_OK) return st;
    return stage_download(ctx, in_bytes, d_out, out_row, height, dst, dst_step, stream);
}

}  // extern "C"
