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
//   k_lv_final   pass 3: cv::adaptiveThreshold(G, 127, MEAN_C, BINARY, 15, 0) through a 46 x 22 LDS tile, then
//                ((A - N) > minResultVariance)
// Bound: HBM for passes 1-2 (3 B/px read each, 2 B/px written), LDS for pass 3.  WithoutFilters is integer-exact against
// the oracle; the filtered variant is compared under a tolerance (float32 log / exp / pow differ between libraries).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "prl_internal.h"

namespace prl_hip {
namespace {

constexpr int TX = 32, TY = 8;  // pixels per workgroup (256 threads)

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

// variance of the 3 x 3 neighbourhood of tile position (ty, tx) for channel c; `t` holds the replicate-clamped page
// pixels of the tile with a border of B >= 1 (row pitch P, 3 bytes per pixel)
template <int P>
__device__ __forceinline__ float var_at(const uint8_t (*t)[P], int ty, int tx, int c)
{
    int s = 0, q = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int p = t[ty + dy][(tx + dx) * 3 + c];
            s += p;
            q += p * p;
        }
    const float scale = (float)(1 / (9.0f * 9.0f));
    const float v = (float)(9 * q - s * s) * scale;
    return v > 0.01f ? v : 0.01f;
}

template <int B>
__device__ __forceinline__ void load_tile(const uint8_t* page, size_t step, int width, int height, int x0, int y0,
                                          uint8_t (*t)[(TX + 2 * B) * 3])
{
    constexpr int W3 = (TX + 2 * B) * 3, H = TY + 2 * B;
    for (int i = threadIdx.x; i < W3 * H; i += 256) {
        const int r = i / W3, b = i - r * W3, px = b / 3, c = b - px * 3;
        t[r][b] = page[(size_t)clampi(y0 - B + r, 0, height - 1) * step + (size_t)clampi(x0 - B + px, 0, width - 1) * 3 + c];
    }
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
// words, 13.5 ms for 16 pages of which 13 were the atomics).
__global__ void __launch_bounds__(256) k_lv_stats(PageSet src, int width, int height, LvStats* __restrict__ stats)
{
    __shared__ uint8_t tile[TY + 2][(TX + 2) * 3];
    __shared__ unsigned r_min[4], r_max[4];
    __shared__ double r_sum;
    const int page = blockIdx.y;
    const int tiles_x = (width + TX - 1) / TX, tiles_y = (height + TY - 1) / TY;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    unsigned mn[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[4] = {0u, 0u, 0u, 0u};
    double s = 0.0;
    for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
        const int x0 = (t % tiles_x) * TX, y0 = (t / tiles_x) * TY;
        __syncthreads();  // the previous tile has been consumed
        load_tile<1>(src.page(page), src.step, width, height, x0, y0, tile);
        __syncthreads();
        if (x0 + tx < width && y0 + ty < height) {
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v[c] = var_at(tile, ty + 1, tx + 1, c);
                const unsigned o = f2ord(v[c]);
                mn[c] = min(mn[c], o);
                mx[c] = max(mx[c], o);
            }
            const float l = (logf(v[0]) + logf(v[1])) + logf(v[2]);
            const unsigned ol = f2ord(l);
            mn[3] = min(mn[3], ol);
            mx[3] = max(mx[3], ol);
            s += (double)l;
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
    __shared__ uint8_t tile[TY + 4][(TX + 4) * 3];
    __shared__ float var[3][TY + 2][TX + 2];
    const int page = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    load_tile<2>(src.page(page), src.step, p.width, p.height, x0, y0, tile);
    __syncthreads();
    // variance at the tile's pixels and their 1-pixel ring; ring positions outside the page take the clamped pixel's value
    // (BORDER_REPLICATE of the variance map = the variance at the clamped coordinates)
    for (int i = threadIdx.x; i < (TY + 2) * (TX + 2); i += 256) {
        const int r = i / (TX + 2), cx = i - r * (TX + 2);
        const int gy = clampi(y0 - 1 + r, 0, p.height - 1), gx = clampi(x0 - 1 + cx, 0, p.width - 1);
        const int ty = gy - (y0 - 2), tx = gx - (x0 - 2);  // the clamped pixel's position inside the tile
#pragma unroll
        for (int c = 0; c < 3; ++c) var[c][r][cx] = var_at(tile, ty, tx, c);
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= p.width || y >= p.height) return;
    const LvConsts cs = consts[page];
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
        return;
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

__global__ void __launch_bounds__(256) k_lv_final(LvParams p, const uint8_t* __restrict__ G, const uint8_t* __restrict__ NR, size_t plane,
                                                  PageSetOut dst)
{
    __shared__ uint8_t g[TY + 14][TX + 14];
    __shared__ unsigned short hs[TY + 14][TX];
    const int page = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const uint8_t* gp = G + (size_t)page * plane;
    for (int i = threadIdx.x; i < (TY + 14) * (TX + 14); i += 256) {
        const int r = i / (TX + 14), c = i - r * (TX + 14);
        g[r][c] = gp[(size_t)clampi(y0 - 7 + r, 0, p.height - 1) * p.width + clampi(x0 - 7 + c, 0, p.width - 1)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (TY + 14) * TX; i += 256) {
        const int r = i / TX, c = i - r * TX;
        unsigned s = 0;
#pragma unroll
        for (int d = 0; d < 15; ++d) s += g[r][c + d];
        hs[r][c] = (unsigned short)s;
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= p.width || y >= p.height) return;
    unsigned bs = 0;
#pragma unroll
    for (int d = 0; d < 15; ++d) bs += hs[ty + d][tx];
    const int mean = __double2int_rn((double)bs * (1. / 225));
    const int a = ((int)g[ty + 7][tx + 7] - mean > 0) ? 127 : 0;
    const int diff = max(0, a - (int)NR[(size_t)page * plane + (size_t)y * p.width + x]);
    dst.page(page)[(size_t)y * dst.step + x] = diff > p.min_result_variance ? 255 : 0;
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
    if ((height + TY - 1) / TY > 65535) return PRL_ERR_BAD_ARG;
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
        const dim3 grid((unsigned)((width + TX - 1) / TX), (unsigned)((height + TY - 1) / TY), (unsigned)cnt);
        hipLaunchKernelGGL(k_lv_init, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, hs, d_stats, cnt);
        const int tiles = (int)(grid.x * grid.y);
        const int per_page = std::max(1, std::min(tiles, std::max(8, 8192 / cnt)));  // workgroups per page: a few thousand in all
        hipLaunchKernelGGL(k_lv_stats, dim3((unsigned)per_page, (unsigned)cnt), dim3(256), 0, hs, s, width, height, d_stats);
        hipLaunchKernelGGL(k_lv_consts, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, hs, d_stats, d_consts, cnt, width, height, coeff);
        hipLaunchKernelGGL(k_lv_maps, grid, dim3(256), 0, hs, s, p, d_consts, G, NR, plane, d);
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
    st = stage_upload(ctx, 0, src, src_step, in_row, height, d_in, stream);
    if (st != PRL_OK) return st;
    st = prl_hip_binarize_lv_batch_device(1, with_filters, coeff, min_result_variance, gamma, d_in, in_bytes, in_row, width, height,
                                          d_out, out_bytes, out_row, stream);
    if (st != PRL_OK) return st;
    return stage_download(ctx, in_bytes, d_out, out_row, height, dst, dst_step, stream);
}

}  // extern "C"
