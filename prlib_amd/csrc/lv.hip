// lv.hip — prl::binarizeByLocalVariances / prl::binarizeByLocalVariancesWithoutFilters on 8UC3
// pages resident in device memory.
//
// Reference: src/binarizations/binarizeByLocalVariances.cpp:13-145 (with filters), :148-292 (without),
// src/imageLibCommon.cpp:397-466 (MatToLocalVarianceMap).  The OpenCV arithmetic is restated, with the assumptions it
// needs (float32 filter2D accumulation order, cv::log / cv::exp replaced by logf / expf), in the test oracle's
// local-variance file; this file follows it operation by operation.
//
// The float32 variance map (12 B per pixel) is never stored: every pass rebuilds it from the 8-bit page, whose 3 x 3 sums
// are exact integers (< 2^24), through an LDS tile.
//   k_lv_stats   pass 1: per page min / max of the variance per channel, min / max / sum of the log map (block reduction,
//                ordered-integer atomics for the float extrema, one float64 atomic per block for the sum)
//   k_lv_maps    pass 2: result1 & result2 (variance > 10; contrast-filtered variance above the per-channel threshold) and,
//                with filters, the gamma-corrected 8-bit log map G and the 8-bit noise term N (N = 255 where
//                result1 & result2 is false, which zeroes the pixel in pass 3); without filters the final mask
//   k_lv_final   pass 3: cv::adaptiveThreshold(G, 127, MEAN_C, BINARY, 15, 0) through a 78 x 46 LDS tile (sliding box sums), then
//                ((A - N) > minResultVariance)
// Bound: vector issue and LDS in all three passes (profiles/r02/pmc_stages.txt), not HBM.  WithoutFilters is integer-exact against
// the oracle; the filtered variant is compared under a tolerance (float32 log / exp / pow differ between libraries).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "prl_internal.h"

namespace prl_hip {
namespace {


struct LvStats {
    unsigned vmin[3], vmax[3];  // ordered-integer images of the float extrema
    unsigned lmin, lmax;
    double lsum;
};

__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Tile of the variance passes: MX x MY pixels per workgroup (4 per thread) plus a border of B.  The page bytes go to LDS once
// (whole dwords where the tile lies inside the page), then the HORIZONTAL 3-sums of p and p^2 of every position are packed into
// one word (sum <= 765 in bits 0..11, sum of squares <= 195075 from bit 12; three of them still fit: 2295, 585225), so that a
// variance is three LDS words and two additions instead of nine bytes, nine additions and nine multiply-adds.
// (First version: 32 x 8 tiles, 27 byte reads per pixel: 91 vector instructions per channel-pixel, profiles/r02/pmc_stages.txt.)
constexpr int MX = 64, MY = 16;
template <int B>
struct TileGeo {
    static constexpr int PW = MX + 2 * B, PH = MY + 2 * B;  // staged pixels
    static constexpr int PITCH = (PW * 3 + 3) / 4 * 4;      // bytes per staged row
    static constexpr int HW3 = (PW - 2) * 3;                // packed sums per row: pixel columns 1 .. PW-2, three channels
};

template <int B>
__device__ __forceinline__ void stage_tile(const uint8_t* page, size_t step, int width, int height, int x0, int y0,
                                           uint8_t (*t)[TileGeo<B>::PITCH], unsigned (*H)[TileGeo<B>::HW3])
{
    using G = TileGeo<B>;
    constexpr int DW = G::PITCH / 4;
    const bool inside = x0 - B >= 0 && y0 - B >= 0 && x0 + MX + B + 2 <= width && y0 + MY + B <= height;
    if (inside) {
        for (int i = threadIdx.x; i < G::PH * DW; i += 256) {
            const int r = i / DW, k = i - r * DW;
            unsigned v;
            __builtin_memcpy(&v, page + (size_t)(y0 - B + r) * step + (size_t)(x0 - B) * 3 + 4 * k, 4);
            reinterpret_cast<unsigned*>(t[r])[k] = v;
        }
    } else {
        for (int i = threadIdx.x; i < G::PH * G::PW * 3; i += 256) {
            const int r = i / (G::PW * 3), b = i - r * (G::PW * 3), px = b / 3, c = b - px * 3;
            t[r][b] = page[(size_t)clampi(y0 - B + r, 0, height - 1) * step + (size_t)clampi(x0 - B + px, 0, width - 1) * 3 + c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G::PH * G::HW3; i += 256) {
        const int r = i / G::HW3, b = i - r * G::HW3;
        const unsigned p0 = t[r][b], p1 = t[r][b + 3], p2 = t[r][b + 6];
        H[r][b] = (p0 + p1 + p2) | ((p0 * p0 + p1 * p1 + p2 * p2) << 12);
    }
    __syncthreads();
}

// variance of the 3 x 3 neighbourhood of staged pixel (ty, tx), channel c (MatToLocalVarianceMap, imageLibCommon.cpp:447-465)
template <int HW3>
__device__ __forceinline__ float var_h(const unsigned (*H)[HW3], int ty, int tx, int c)
{
    const int b = (tx - 1) * 3 + c;
    const unsigned v3 = H[ty - 1][b] + H[ty][b] + H[ty + 1][b];
    const int s = (int)(v3 & 0xfffu), q = (int)(v3 >> 12);
    const float scale = (float)(1 / (9.0f * 9.0f));
    const float v = (float)(9 * q - s * s) * scale;
    return v > 0.01f ? v : 0.01f;
}

__global__ void __launch_bounds__(256) k_lv_init(LvStats* st, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < 3; ++c) { st[i].vmin[c] = 0xffffffffu; st[i].vmax[c] = 0u; }
    st[i].lmin = 0xffffffffu;
    st[i].lmax = 0u;
    st[i].lsum = 0.0;
}

// Each workgroup walks a share of the page's tiles and keeps running extrema / a running sum in registers; one set of
// atomics per WORKGROUP at the end (the first version issued them per tile: 42 000 workgroups per A4 page on the same nine
// words, 13.5 ms for 16 pages of which 13 were the atomics).  LOGS: the log map's extrema and sum (the filtered variant only).
template <bool LOGS>
__global__ void __launch_bounds__(256) k_lv_stats(PageSet src, int width, int height, LvStats* __restrict__ stats)
{
    using G = TileGeo<1>;
    __shared__ __attribute__((aligned(16))) uint8_t tile[G::PH][G::PITCH];
    __shared__ unsigned H[G::PH][G::HW3];
    __shared__ unsigned r_min[4], r_max[4];
    __shared__ double r_sum;
    const int page = blockIdx.y;
    const int tiles_x = (width + MX - 1) / MX, tiles_y = (height + MY - 1) / MY;
    const int tx = threadIdx.x % MX, ty0 = threadIdx.x / MX;
    unsigned mn[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[4] = {0u, 0u, 0u, 0u};
    double s = 0.0;
    for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
        const int x0 = (t % tiles_x) * MX, y0 = (t / tiles_x) * MY;
        __syncthreads();  // the previous tile has been consumed
        stage_tile<1>(src.page(page), src.step, width, height, x0, y0, tile, H);
#pragma unroll
        for (int k = 0; k < MY / 4; ++k) {
            const int ty = ty0 + 4 * k;
            if (x0 + tx < width && y0 + ty < height) {
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    v[c] = var_h(H, ty + 1, tx + 1, c);
                    const unsigned o = f2ord(v[c]);
                    mn[c] = min(mn[c], o);
                    mx[c] = max(mx[c], o);
                }
                if (LOGS) {
                    const float l = (logf(v[0]) + logf(v[1])) + logf(v[2]);
                    const unsigned ol = f2ord(l);
                    mn[3] = min(mn[3], ol);
                    mx[3] = max(mx[3], ol);
                    s += (double)l;
                }
            }
        }
    }
    if (threadIdx.x < 4) { r_min[threadIdx.x] = 0xffffffffu; r_max[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) r_sum = 0.0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mn[c] = min(mn[c], (unsigned)__shfl_xor((int)mn[c], o));
            mx[c] = max(mx[c], (unsigned)__shfl_xor((int)mx[c], o));
        }
        s += __shfl_xor(s, o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { atomicMin(&r_min[c], mn[c]); atomicMax(&r_max[c], mx[c]); }
        atomicAdd(&r_sum, s);
    }
    __syncthreads();
    LvStats* st = stats + page;
    if (threadIdx.x < 3) { atomicMin(&st->vmin[threadIdx.x], r_min[threadIdx.x]); atomicMax(&st->vmax[threadIdx.x], r_max[threadIdx.x]); }
    if (threadIdx.x == 3) { atomicMin(&st->lmin, r_min[3]); atomicMax(&st->lmax, r_max[3]); atomicAdd(&st->lsum, r_sum); }
}

// Per-page constants of the later passes, once per page instead of once per pixel (two float64 divisions among them).
struct LvConsts {
    float thr[3];      // ((max - min) / 2) * coeff per channel      binarizeByLocalVariances.cpp:83-85
    float ga, gb;      // the convertTo scale / offset of the log map  :116-119
    float lmean;       // cv::mean of the log map                      :131
    float pad[2];
};

__global__ void k_lv_consts(const LvStats* __restrict__ stats, LvConsts* __restrict__ out, int n, int width, int height, double coeff)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LvStats st = stats[i];
    LvConsts c;
    for (int k = 0; k < 3; ++k) {
        const float dist = ord2f(st.vmax[k]) - ord2f(st.vmin[k]);
        const float half = (float)((double)dist * (1. / 2));
        c.thr[k] = (float)((double)half * coeff);
    }
    const float lmin = ord2f(st.lmin), lmax = ord2f(st.lmax);
    const double range = (double)lmax - (double)lmin;
    c.ga = (float)(1.0 / range);
    c.gb = (float)(-(double)lmin / range);
    c.lmean = (float)(st.lsum / ((double)width * (double)height));
    c.pad[0] = c.pad[1] = 0.0f;
    out[i] = c;
}

struct LvParams {
    int width, height, with_filters, min_result_variance;
    double coeff, gamma;
};

__global__ void __launch_bounds__(256) k_lv_maps(PageSet src, LvParams p, const LvConsts* __restrict__ consts, uint8_t* __restrict__ G,
                                                 uint8_t* __restrict__ NR, size_t plane, PageSetOut dst)
{
    using Geo = TileGeo<2>;
    __shared__ __attribute__((aligned(16))) uint8_t tile[Geo::PH][Geo::PITCH];
    __shared__ unsigned H[Geo::PH][Geo::HW3];
    __shared__ float var[3][MY + 2][MX + 2];
    const int page = blockIdx.z, x0 = blockIdx.x * MX, y0 = blockIdx.y * MY;
    stage_tile<2>(src.page(page), src.step, p.width, p.height, x0, y0, tile, H);
    // variance at the tile's pixels and their 1-pixel ring; ring positions outside the page take the clamped pixel's value
    // (BORDER_REPLICATE of the variance map = the variance at the clamped coordinates)
    for (int i = threadIdx.x; i < (MY + 2) * (MX + 2); i += 256) {
        const int r = i / (MX + 2), cx = i - r * (MX + 2);
        const int gy = clampi(y0 - 1 + r, 0, p.height - 1), gx = clampi(x0 - 1 + cx, 0, p.width - 1);
        const int ty = gy - (y0 - 2), tx = gx - (x0 - 2);  // the clamped pixel's position inside the staged tile
#pragma unroll
        for (int c = 0; c < 3; ++c) var[c][r][cx] = var_h(H, ty, tx, c);
    }
    __syncthreads();
    const int tx = threadIdx.x % MX, ty0 = threadIdx.x / MX;
    const int x = x0 + tx;
    if (x >= p.width) return;
    const LvConsts cs = consts[page];
#pragma unroll 1
    for (int k = 0; k < MY / 4; ++k) {
        const int ty = ty0 + 4 * k, y = y0 + ty;
        if (y >= p.height) break;
        bool r1 = false, r2 = false;
        float vc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float thr = cs.thr[c];
            const float up = var[c][ty][tx + 1], lf = var[c][ty + 1][tx], ce = var[c][ty + 1][tx + 1], rt = var[c][ty + 1][tx + 2],
                        dn = var[c][ty + 2][tx + 1];
            vc[c] = ce;
            float s;
            if (p.with_filters) {  // cv::filter2D: the five non-zero taps in row-major order
                s = 0.0f;
                s += -1.0f * up;
                s += -1.0f * lf;
                s += 16.0f * ce;
                s += -1.0f * rt;
                s += -1.0f * dn;
                r1 = r1 || ce > 10.0f;
            } else {               // the reference's own nine-term Vec3f sum, zeros included
                const float ul = var[c][ty][tx], ur = var[c][ty][tx + 2], dl = var[c][ty + 2][tx], dr = var[c][ty + 2][tx + 2];
                s = ul * 0.0f;
                s = s + up * -1.0f;
                s = s + ur * 0.0f;
                s = s + lf * -1.0f;
                s = s + ce * 16.0f;
                s = s + rt * -1.0f;
                s = s + dl * 0.0f;
                s = s + dn * -1.0f;
                s = s + dr * 0.0f;
            }
            r2 = r2 || s > thr;
        }
        if (!p.with_filters) {
            float mv = vc[0] > vc[1] ? vc[0] : vc[1];
            mv = mv > vc[2] ? mv : vc[2];
            r1 = mv > (float)p.min_result_variance;
            dst.page(page)[(size_t)y * dst.step + x] = (r1 && r2) ? 255 : 0;
            continue;
        }
        const float ga = cs.ga, gb = cs.gb, lmean = cs.lmean;
        const float l = (logf(vc[0]) + logf(vc[1])) + logf(vc[2]);
        const float t = l * ga + gb;
        const float tg = p.gamma == 2.0 ? t * t : powf(t, (float)p.gamma);
        const float g255 = rintf(fabsf(tg * 255.0f));
        const float d = l - lmean;
        const float e = expf(-(d * d) * 0.5f);
        const float n127 = rintf(fabsf(e * 127.0f));
        const size_t o = (size_t)page * plane + (size_t)y * p.width + x;
        G[o] = (uint8_t)(g255 > 255.0f ? 255.0f : g255);   // (NaN - a page whose log map is constant - compares false: 0, as saturate_cast does)
        NR[o] = (r1 && r2) ? (uint8_t)(n127 > 255.0f ? 255.0f : n127) : (uint8_t)255;
    }
}

// Pass 3: cv::adaptiveThreshold(G, 127, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY, 15, 0) + the noise term.  FX x FY outputs per
// workgroup from a (FX + 14) x (FY + 14) tile of G (replicate border); the 15-tap box sums slide: a thread forms the first sum
// of its run of eight positions and then adds the entering and subtracts the leaving value, horizontally (rows of the tile)
// and then vertically (columns of the row sums) - 3.6 operations per position and direction instead of 15.
constexpr int FX = 64, FY = 32;
__global__ void __launch_bounds__(256) k_lv_final(LvParams p, const uint8_t* __restrict__ G, const uint8_t* __restrict__ NR, size_t plane,
                                                  PageSetOut dst)
{
    constexpr int GW = FX + 14, GH = FY + 14, GP = (GW + 3) / 4 * 4;
    __shared__ uint8_t g[GH][GP];
    __shared__ unsigned short hs[GH][FX];
    const int page = blockIdx.z, x0 = blockIdx.x * FX, y0 = blockIdx.y * FY;
    const uint8_t* gp = G + (size_t)page * plane;
    for (int i = threadIdx.x; i < GH * GW; i += 256) {
        const int r = i / GW, c = i - r * GW;
        g[r][c] = gp[(size_t)clampi(y0 - 7 + r, 0, p.height - 1) * p.width + clampi(x0 - 7 + c, 0, p.width - 1)];
    }
    __syncthreads();
    // horizontal: (row, run of 8 columns) per step
    for (int i = threadIdx.x; i < GH * (FX / 8); i += 256) {
        const int r = i / (FX / 8), c0 = (i - r * (FX / 8)) * 8;
        unsigned s = 0;
#pragma unroll
        for (int d = 0; d < 15; ++d) s += g[r][c0 + d];
        hs[r][c0] = (unsigned short)s;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            s += g[r][c0 + k + 14];
            s -= g[r][c0 + k - 1];
            hs[r][c0 + k] = (unsigned short)s;
        }
    }
    __syncthreads();
    // vertical: thread = (column, run of 8 rows)
    const int tx = threadIdx.x % FX, ty0 = (threadIdx.x / FX) * 8;
    const int x = x0 + tx;
    if (x >= p.width) return;
    unsigned bs = 0;
#pragma unroll
    for (int d = 0; d < 15; ++d) bs += hs[ty0 + d][tx];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ty = ty0 + k, y = y0 + ty;
        if (k > 0) {
            bs += hs[ty + 14][tx];
            bs -= hs[ty - 1][tx];
        }
        if (y < p.height) {
            const int mean = __double2int_rn((double)bs * (1. / 225));
            const int a = ((int)g[ty + 7][tx + 7] - mean > 0) ? 127 : 0;
            const int diff = max(0, a - (int)NR[(size_t)page * plane + (size_t)y * p.width + x]);
            dst.page(page)[(size_t)y * dst.step + x] = diff > p.min_result_variance ? 255 : 0;
        }
    }
}

}  // namespace
}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

int prl_hip_binarize_lv_batch_device(int n_pages, int with_filters, double coeff, int min_result_variance, double gamma,
                                     const uint8_t* d_src, size_t src_page_stride, size_t src_step, int width, int height,
                                     uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;  // binarizeByLocalVariances.cpp:16-19, :151-154
    if (n_pages < 0 || !d_src || !d_dst || src_step < (size_t)width * 3 || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    if ((height + MY - 1) / MY > 65535) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t plane = ((size_t)width * height + 255) / 256 * 256;
    const int chunk = std::min(n_pages, 16384);
    const size_t stats_bytes = ((size_t)chunk * (sizeof(LvStats) + sizeof(LvConsts)) + 255) / 256 * 256;
    st = ensure_scratch(ctx, stats_bytes + (with_filters ? 2 * plane * (size_t)chunk : 0));
    if (st != PRL_OK) return st;
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(hs, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    LvStats* d_stats = static_cast<LvStats*>(ctx->scratch);
    LvConsts* d_consts = reinterpret_cast<LvConsts*>(d_stats + chunk);
    uint8_t* G = static_cast<uint8_t*>(ctx->scratch) + stats_bytes;
    uint8_t* NR = G + plane * (size_t)chunk;
    LvParams p{width, height, with_filters ? 1 : 0, min_result_variance, coeff, gamma};
    for (int first = 0; first < n_pages; first += chunk) {
        const int cnt = std::min(chunk, n_pages - first);
        PageSet s{};
        s.base = d_src + (size_t)first * src_page_stride; s.page_stride = src_page_stride; s.step = src_step;
        PageSetOut d{};
        d.base = d_dst + (size_t)first * dst_page_stride; d.page_stride = dst_page_stride; d.step = dst_step;
        const dim3 grid((unsigned)((width + FX - 1) / FX), (unsigned)((height + FY - 1) / FY), (unsigned)cnt);    // k_lv_final
        const dim3 mgrid((unsigned)((width + MX - 1) / MX), (unsigned)((height + MY - 1) / MY), (unsigned)cnt);   // variance passes
        hipLaunchKernelGGL(k_lv_init, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, hs, d_stats, cnt);
        const int tiles = (int)(mgrid.x * mgrid.y);
        const int per_page = std::max(1, std::min(tiles, std::max(8, 8192 / cnt)));  // workgroups per page: a few thousand in all
        if (with_filters) hipLaunchKernelGGL(k_lv_stats<true>, dim3((unsigned)per_page, (unsigned)cnt), dim3(256), 0, hs, s, width, height, d_stats);
        else hipLaunchKernelGGL(k_lv_stats<false>, dim3((unsigned)per_page, (unsigned)cnt), dim3(256), 0, hs, s, width, height, d_stats);
        hipLaunchKernelGGL(k_lv_consts, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, hs, d_stats, d_consts, cnt, width, height, coeff);
        hipLaunchKernelGGL(k_lv_maps, mgrid, dim3(256), 0, hs, s, p, d_consts, G, NR, plane, d);
        if (with_filters) hipLaunchKernelGGL(k_lv_final, grid, dim3(256), 0, hs, p, G, NR, plane, d);
        PRL_HIP_CHECK(hipGetLastError());
    }
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, hs));
    return PRL_OK;
}

int prl_hip_binarize_lv_host(int with_filters, double coeff, int min_result_variance, double gamma, const uint8_t* src,
                             size_t src_step, int width, int height, uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY;
    if (!dst || src_step < (size_t)width * 3 || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    const size_t in_row = (size_t)width * 3, out_row = (size_t)width;
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
    st = prl_hip_binarize_lv_batch_device(1, with_filters, coeff, min_result_variance, gamma, d_in, in_bytes, in_row, width, height,
                                          d_out, out_bytes, out_row, stream);
    if (st != PRL_OK) return st;
    return stage_download(ctx, in_bytes, d_out, out_row, height, dst, dst_step, stream);
}

}  // extern "C"
