// glue.hip — SURVEY.md §8f rank 2: the channel adapters and the device-resident hand-off between the hot-path
// stages, so a page goes denoise -> gray -> binarize -> (invert) -> thinning without a host round trip.
//
//   k_bgr2gray : cv::cvtColor(COLOR_BGR2GRAY / BGRA2GRAY) on 8-bit data, the call every binarizer makes first for a
//                colour input (src/binarizations/binarizeSauvola.cpp:51; same line in Niblack/Wolf/NICK/Feng and in
//                src/thinning/thinZhangSuen.cpp:78): 14-bit fixed point (1868 B + 9617 G + 4899 R + 8192) >> 14
//                [upstream, SURVEY.md Appendix B], integer => bit-exact.
//   k_gray2bgr : cv::cvtColor(COLOR_GRAY2BGR): the replicate a user needs in front of prl::denoise, which only
//                accepts 3/4-channel input.
//   k_invert   : cv::bitwise_not: the binarizers emit white = background, thinning thins white (§3.4).
// All three are pure byte streams (HBM-bound: 4, 4 and 2 B per pixel); a thread handles 4 pixels with dword
// accesses when the rows are 4-byte aligned, byte accesses otherwise.
//
// prl_hip_chain_batch_device strings the public entry points together on one stream with its intermediates in the
// device staging workspace.  There is no such function in the reference (a user writes the calls one after the
// other, each through host memory); BASELINE config 5 is this chain.
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <thread>
#include <string>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {
namespace {

__device__ __forceinline__ unsigned gray14(unsigned b, unsigned g, unsigned r)
{
    return (b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14;
}

template <int CH>
__global__ void __launch_bounds__(256) k_bgr2gray(PageSet src, PageSetOut dst, int width, int height)
{
    const int page = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= width) return;
    const uint8_t* s = src.page(page) + (size_t)y * src.step + (size_t)x0 * CH;
    uint8_t* d = dst.page(page) + (size_t)y * dst.step + x0;
    const int n = min(4, width - x0);
    unsigned g[4] = {0, 0, 0, 0};
    if (n == 4 && (((size_t)s) & 3) == 0) {
        const unsigned* q = reinterpret_cast<const unsigned*>(s);
        if (CH == 3) {
            const unsigned w0 = q[0], w1 = q[1], w2 = q[2];  // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            g[0] = gray14(w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff);
            g[1] = gray14(w0 >> 24, w1 & 0xff, (w1 >> 8) & 0xff);
            g[2] = gray14((w1 >> 16) & 0xff, w1 >> 24, w2 & 0xff);
            g[3] = gray14((w2 >> 8) & 0xff, (w2 >> 16) & 0xff, w2 >> 24);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = gray14(q[i] & 0xff, (q[i] >> 8) & 0xff, (q[i] >> 16) & 0xff);
        }
    } else {
        for (int i = 0; i < n; ++i) g[i] = gray14(s[i * CH], s[i * CH + 1], s[i * CH + 2]);
    }
    if (n == 4 && (((size_t)d) & 3) == 0) {
        *reinterpret_cast<unsigned*>(d) = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
    } else {
        for (int i = 0; i < n; ++i) d[i] = (uint8_t)g[i];
    }
}

template <int CH>
__global__ void __launch_bounds__(256) k_gray2bgr(PageSet src, PageSetOut dst, int width, int height)
{
    const int page = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= width) return;
    const uint8_t* s = src.page(page) + (size_t)y * src.step + x0;
    uint8_t* d = dst.page(page) + (size_t)y * dst.step + (size_t)x0 * CH;
    const int n = min(4, width - x0);
    unsigned g[4] = {0, 0, 0, 0};
    if (n == 4 && (((size_t)s) & 3) == 0) {
        const unsigned w = *reinterpret_cast<const unsigned*>(s);
        g[0] = w & 0xff; g[1] = (w >> 8) & 0xff; g[2] = (w >> 16) & 0xff; g[3] = w >> 24;
    } else {
        for (int i = 0; i < n; ++i) g[i] = s[i];
    }
    if (n == 4 && (((size_t)d) & 3) == 0) {
        unsigned* q = reinterpret_cast<unsigned*>(d);
        if (CH == 3) {
            q[0] = g[0] * 0x00010101u | (g[1] << 24);
            q[1] = g[1] * 0x00000101u | (g[2] * 0x01010000u);
            q[2] = g[2] | (g[3] * 0x01010100u);
        } else {  // alpha = 255 (cv::cvtColor GRAY2BGRA)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = g[i] * 0x00010101u | 0xff000000u;
        }
    } else {
        for (int i = 0; i < n; ++i) {
            d[i * CH] = d[i * CH + 1] = d[i * CH + 2] = (uint8_t)g[i];
            if (CH == 4) d[i * CH + 3] = 255;
        }
    }
}

__global__ void __launch_bounds__(256) k_invert(PageSet src, PageSetOut dst, int width, int height)
{
    const int page = blockIdx.z, y = blockIdx.y;
    const uint8_t* s = src.page(page) + (size_t)y * src.step;
    uint8_t* d = dst.page(page) + (size_t)y * dst.step;
    // align on the destination: head bytes, then 16-byte stores (source fetched as dwords / bytes when it is not
    // co-aligned), tail bytes
    const int head = min(width, (int)((16 - ((size_t)d & 15)) & 15));
    if (blockIdx.x == 0 && (int)threadIdx.x < head) d[threadIdx.x] = (uint8_t)~s[threadIdx.x];
    const int x0 = head + (blockIdx.x * 256 + threadIdx.x) * 16;
    if (x0 >= width) return;
    if (x0 + 16 <= width) {
        uint4 w;
        const size_t sa = (size_t)(s + x0);
        if ((sa & 15) == 0) {
            w = *reinterpret_cast<const uint4*>(s + x0);
        } else if ((sa & 3) == 0) {
            const unsigned* q = reinterpret_cast<const unsigned*>(s + x0);
            w = make_uint4(q[0], q[1], q[2], q[3]);
        } else {  // aligned dwords + byte funnel
            const unsigned sh = (unsigned)(sa & 3);
            const unsigned* q = reinterpret_cast<const unsigned*>(s + x0 - sh);
            const unsigned a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];  // q[4] holds bytes x0+16-sh.. : inside the row
            w = make_uint4(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh),
                           __builtin_amdgcn_alignbyte(a3, a2, sh), __builtin_amdgcn_alignbyte(a4, a3, sh));
        }
        *reinterpret_cast<uint4*>(d + x0) = make_uint4(~w.x, ~w.y, ~w.z, ~w.w);
    } else {
        for (int i = x0; i < width; ++i) d[i] = (uint8_t)~s[i];
    }
}

struct Args {
    PageSet ps{};
    PageSetOut pd{};
    dim3 grid;
};

int common_args(int n_pages, const uint8_t* d_src, size_t sps, size_t sstep, size_t src_row_bytes, int width, int height,
                uint8_t* d_dst, size_t dps, size_t dstep, size_t dst_row_bytes, Args* a)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (n_pages < 0 || !d_src || !d_dst || sstep < src_row_bytes || dstep < dst_row_bytes) return PRL_ERR_BAD_ARG;
    if (height > 65535) return PRL_ERR_BAD_ARG;  // grid.y limit (one grid row per image row)
    a->ps.base = d_src; a->ps.page_stride = sps; a->ps.step = sstep;
    a->pd.base = d_dst; a->pd.page_stride = dps; a->pd.step = dstep;
    a->grid = dim3((unsigned)((width + 1023) / 1024), (unsigned)height, (unsigned)n_pages);  // 4 px per thread
    return PRL_OK;
}

}  // namespace
}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

int prl_hip_bgr2gray_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                  int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                  void* stream)
{
    if (channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (n_pages > 32768 && d_src && d_dst) {  // grid.z holds at most 65535 pages
        for (int first = 0; first < n_pages; first += 32768) {
            const int st2 = prl_hip_bgr2gray_batch_device(std::min(32768, n_pages - first), channels, d_src + (size_t)first * src_page_stride,
                            src_page_stride, src_step, width, height, d_dst + (size_t)first * dst_page_stride, dst_page_stride,
                            dst_step, stream);
            if (st2 != PRL_OK) return st2;
        }
        return PRL_OK;
    }
    Args a;
    int st = common_args(n_pages, d_src, src_page_stride, src_step, (size_t)(width > 0 ? width : 0) * channels, width, height,
                         d_dst, dst_page_stride, dst_step, (size_t)(width > 0 ? width : 0), &a);
    if (st != PRL_OK || n_pages == 0) return st;
    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3) hipLaunchKernelGGL(k_bgr2gray<3>, a.grid, dim3(256), 0, s, a.ps, a.pd, width, height);
    else hipLaunchKernelGGL(k_bgr2gray<4>, a.grid, dim3(256), 0, s, a.ps, a.pd, width, height);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

int prl_hip_gray2bgr_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                  int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                  void* stream)
{
    if (channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (n_pages > 32768 && d_src && d_dst) {  // grid.z holds at most 65535 pages
        for (int first = 0; first < n_pages; first += 32768) {
            const int st2 = prl_hip_gray2bgr_batch_device(std::min(32768, n_pages - first), channels, d_src + (size_t)first * src_page_stride,
                            src_page_stride, src_step, width, height, d_dst + (size_t)first * dst_page_stride, dst_page_stride,
                            dst_step, stream);
            if (st2 != PRL_OK) return st2;
        }
        return PRL_OK;
    }
    Args a;
    int st = common_args(n_pages, d_src, src_page_stride, src_step, (size_t)(width > 0 ? width : 0), width, height, d_dst,
                         dst_page_stride, dst_step, (size_t)(width > 0 ? width : 0) * channels, &a);
    if (st != PRL_OK || n_pages == 0) return st;
    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3) hipLaunchKernelGGL(k_gray2bgr<3>, a.grid, dim3(256), 0, s, a.ps, a.pd, width, height);
    else hipLaunchKernelGGL(k_gray2bgr<4>, a.grid, dim3(256), 0, s, a.ps, a.pd, width, height);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

int prl_hip_invert_batch_device(int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step, int width,
                                int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (n_pages > 32768 && d_src && d_dst) {  // grid.z holds at most 65535 pages
        for (int first = 0; first < n_pages; first += 32768) {
            const int st2 = prl_hip_invert_batch_device(std::min(32768, n_pages - first), d_src + (size_t)first * src_page_stride,
                                                        src_page_stride, src_step, width, height,
                                                        d_dst + (size_t)first * dst_page_stride, dst_page_stride, dst_step, stream);
            if (st2 != PRL_OK) return st2;
        }
        return PRL_OK;
    }
    Args a;
    int st = common_args(n_pages, d_src, src_page_stride, src_step, (size_t)(width > 0 ? width : 0), width, height, d_dst,
                         dst_page_stride, dst_step, (size_t)(width > 0 ? width : 0), &a);
    if (st != PRL_OK || n_pages == 0) return st;
    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;
    a.grid.x = (unsigned)((width + 15 + 4095) / 4096);  // 16 px per thread after a head of up to 15
    hipLaunchKernelGGL(k_invert, a.grid, dim3(256), 0, static_cast<hipStream_t>(stream), a.ps, a.pd, width, height);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// [deskew] -> [denoise] -> [backgroundNormalization] -> gray -> binarize -> [thinning of the inverted mask]: BASELINE
// config 5's order.  Every deskewed page has its own size, so the stages after it run per RUN of consecutive pages of
// equal size (all pages of a skewed batch come out max(W,H) square: one run).
namespace {

size_t r256(size_t v) { return (v + 255) / 256 * 256; }

struct ChainLayout {   // per page of a uniform run, each part rounded up to 256 B
    size_t denoised, normalised, gray, mask, total;
    int ch_after;      // channels of the page the gray conversion sees
};

ChainLayout chain_layout(const prl_chain_params* cp, int channels, int width, int height, const prl_binarize_geometry& g)
{
    ChainLayout l{};
    const size_t px = (size_t)width * height;
    int ch = channels;
    if (cp->denoise) l.denoised = r256(px * (size_t)ch);
    if (cp->background_normalization) {
        ch = prl_hip_bgnorm_out_channels(ch);
        l.normalised = r256(px * (size_t)ch);
    }
    l.ch_after = ch;
    if (ch != 1) l.gray = r256(px);
    if (cp->thin != PRL_CHAIN_NO_THINNING) l.mask = r256((size_t)g.out_w * g.out_h);
    l.total = l.denoised + l.normalised + l.gray + l.mask;
    return l;
}

// The stages after deskew on `cnt` pages of one size; ws: cnt * layout.total bytes.  Two halves: chain_uniform_a is the
// denoise stage (NL-means: compute), chain_uniform_b the streaming stages that follow; the chain may run all first halves of
// a pass before its second halves (see the pass loop).  RunState: where the pages stand after the first half.
struct RunState {
    const uint8_t* cur;
    size_t cur_ps, cur_step;
    uint8_t* w;
};

int chain_uniform_a(const prl_chain_params* cp, int cnt, int channels, const uint8_t* src, size_t src_ps, size_t src_step, int width,
                    int height, uint8_t* ws, void* stream, RunState* rs)
{
    prl_binarize_geometry g;
    int st = prl_hip_binarize_geometry(&cp->binarize, width, height, &g);
    if (st != PRL_OK) return st;
    const ChainLayout l = chain_layout(cp, channels, width, height, g);
    rs->cur = src; rs->cur_ps = src_ps; rs->cur_step = src_step; rs->w = ws;
    if (cp->denoise) {
        st = prl_hip_denoise_batch_device(cnt, channels, cp->denoise_strength, src, src_ps, src_step, width, height, ws, l.denoised,
                                          (size_t)width * channels, stream);
        if (st != PRL_OK) return st;
        rs->cur = ws; rs->cur_ps = l.denoised; rs->cur_step = (size_t)width * channels;
        rs->w = ws + l.denoised * (size_t)cnt;
    }
    return PRL_OK;
}

int chain_uniform_b(const prl_chain_params* cp, int cnt, int channels, const RunState& rs, int width, int height, uint8_t* out,
                    size_t dst_ps, size_t dst_step, void* stream)
{
    prl_binarize_geometry g;
    int st = prl_hip_binarize_geometry(&cp->binarize, width, height, &g);
    if (st != PRL_OK) return st;
    const ChainLayout l = chain_layout(cp, channels, width, height, g);
    const uint8_t* cur = rs.cur;
    size_t cur_ps = rs.cur_ps, cur_step = rs.cur_step;
    int ch = channels;
    uint8_t* w = rs.w;
    if (cp->background_normalization) {
        const int och = prl_hip_bgnorm_out_channels(ch);
        st = prl_hip_bgnorm_batch_device(cnt, ch, cur, cur_ps, cur_step, width, height, w, l.normalised, (size_t)width * och, stream);
        if (st != PRL_OK) return st;
        ch = och;
        cur = w; cur_ps = l.normalised; cur_step = (size_t)width * ch;
        w += l.normalised * (size_t)cnt;
    }
    if (ch != 1) {
        st = prl_hip_bgr2gray_batch_device(cnt, ch, cur, cur_ps, cur_step, width, height, w, l.gray, (size_t)width, stream);
        if (st != PRL_OK) return st;
        cur = w; cur_ps = l.gray; cur_step = (size_t)width;
        w += l.gray * (size_t)cnt;
    }
    // The binarizer's source is this chain's own scratch, which the next pass / the next call overwrites: its flag check (and
    // the literal redo of an overflow-flagged page) must happen HERE, whatever prl_hip_set_deferred_completion says - a
    // pending call resolved later would redo the page from overwritten pixels.
    if (cp->thin == PRL_CHAIN_NO_THINNING) {
        st = prl_hip_binarize_batch_device(&cp->binarize, cnt, cur, cur_ps, cur_step, width, height, out, dst_ps, dst_step, stream);
        if (st != PRL_OK) return st;
        return prl_hip_finish(stream);
    }
    st = prl_hip_binarize_batch_device(&cp->binarize, cnt, cur, cur_ps, cur_step, width, height, w, l.mask, (size_t)g.out_w, stream);
    if (st != PRL_OK) return st;
    st = prl_hip_finish(stream);  // (the mask is final before it is thinned)
    if (st != PRL_OK) return st;
    // cv::bitwise_not between the two stages happens inside the thinning's bit packing (no pass of its own)
    return prl_hip::thin_batch_device(cp->thin, cnt, w, l.mask, (size_t)g.out_w, g.out_w, g.out_h, out, dst_ps, dst_step, stream, true);
}

size_t chain_budget()
{
    return env_knobs().chain_work_mb << 20;  // intermediates of one pass (default 48 GiB: 288 GB of HBM, big passes, few launches)
}

int chain_check(const prl_chain_params* cp, int n_pages, int channels, const uint8_t* d_src, size_t src_step, int width, int height,
                uint8_t* d_dst)
{
    if (!cp) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (cp->denoise && channels == 1) return PRL_ERR_BAD_CHANNELS;  // fastNlMeansDenoisingColored asserts 8UC3 / 8UC4
    if (cp->thin != PRL_CHAIN_NO_THINNING && cp->thin != PRL_THIN_ZHANGSUEN && cp->thin != PRL_THIN_GUOHALL) return PRL_ERR_BAD_ARG;
    if (n_pages < 0 || !d_src || !d_dst || src_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    // the angle search walks (x << 16) fixed-point coordinates, packs points as x | y << 16 and the warp's coordinates
    // saturate to short: the same limit prl_hip_deskew_batch_device / rotate / houghp enforce
    if (cp->deskew && std::max(width, height) > 32767) return PRL_ERR_BAD_ARG;
    return PRL_OK;
}

}  // namespace

extern "C++" {
namespace prl_hip {
// Pages per pass of the chain on n_pages pages of one size (workspace budgets of the chain and of the angle search), and the
// workspace bytes per page.  host_batch.hip sizes its device chunks in whole passes with it.
int chain_pass_layout(const prl_chain_params* cp, int n_pages, int channels, int width, int height, int* pass_pages, size_t* per_page_out,
                      size_t* desk_page_out)
{
    const int len = std::max(width, height);
    const int dw = cp->deskew ? len : width, dh = cp->deskew ? len : height;  // largest page the later stages can see
    prl_binarize_geometry gmax;
    int st = prl_hip_binarize_geometry(&cp->binarize, dw, dh, &gmax);
    if (st != PRL_OK) return st;
    const ChainLayout lmax = chain_layout(cp, channels, dw, dh, gmax);
    const size_t desk_page = cp->deskew ? r256((size_t)len * len * channels) : 0;
    const size_t per_page = desk_page + lmax.total;
    int chunk = per_page == 0 ? n_pages : (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pages, chain_budget() / per_page));
    chunk = std::min(chunk, 32768);
    if (cp->deskew) chunk = std::min(chunk, deskew_pages_per_pass(n_pages, width, height));
    *pass_pages = std::max(1, chunk);
    if (per_page_out) *per_page_out = per_page;
    if (desk_page_out) *desk_page_out = desk_page;
    return PRL_OK;
}
}  // namespace prl_hip
}  // extern "C++"

// The chain on pages whose results may differ in size (deskew): out_wh (host, 2 ints per page) receives each page's
// result size; d_dst pages need room for the largest possible result (prl_hip_chain_max_out_size) at dst_step bytes per row.
int prl_hip_chain_pages_device(const prl_chain_params* cp, int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride,
                               size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                               int32_t* out_wh, double* angles, void* stream)
{
    int st = chain_check(cp, n_pages, channels, d_src, src_step, width, height, d_dst);
    if (st != PRL_OK) return st;
    if (!out_wh) return PRL_ERR_BAD_ARG;
    int max_w = 0, max_h = 0;
    st = prl_hip_chain_max_out_size(cp, width, height, &max_w, &max_h);
    if (st != PRL_OK) return st;
    if (dst_step < (size_t)max_w) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> slk(ctx->stage_mu);  // the staging workspace holds the intermediates

    const int len = std::max(width, height);
    size_t per_page = 0, desk_page = 0;
    int chunk = 0;
    st = chain_pass_layout(cp, n_pages, channels, width, height, &chunk, &per_page, &desk_page);
    if (st != PRL_OK) return st;
    // Pass schedule (with deskew).  Passes must be large enough for the angle search to run near its full rate (>= ~128 pages) and,
    // when NL-means runs beside the next pass's search, small enough for it to finish inside that search: the search's time per
    // page grows as passes shrink (it is latency-bound per page), NL-means' does not.  The sizes below are STARTING values (A4
    // colour scans with ~9 % ink: 192 pages per pass hide NL-means completely, 256 leave it 0.3 s past every search); with the
    // head / body / tail split every pass measures how long its NL-means kernels ran past the search beside them and the
    // following searches are sized from that (shrink in proportion; creep up while there is slack), so other page sizes and ink
    // densities find their own balance.  PRL_HIP_CHAIN_PASS fixes the size (no adaptation).  Without denoise: 256, fixed.
    int first_sz = chunk, main_sz = chunk;
    bool adaptive = false;
    if (cp->deskew && env_knobs().chain_overlap) {
        const int want_main = env_knobs().chain_pass > 0 ? env_knobs().chain_pass : (cp->denoise ? 192 : 256);
        main_sz = std::min(chunk, want_main);
        const int want_first = env_knobs().chain_first_pass > 0 ? env_knobs().chain_first_pass : main_sz;
        first_sz = n_pages >= 3 * want_first ? std::min(main_sz, want_first) : main_sz;
        adaptive = cp->denoise && env_knobs().chain_overlap == 2 && env_knobs().chain_pass == 0;
    }
    // Tail-aware passes (round 5).  The search of a pass takes max(its heaviest page's own time, the pass's share of the chip's
    // atomic rate): kTailS seconds per point for one page (one CU's path for scattered returning atomics), kRateS per point for
    // the chip.  On text scans the second term rules and passes of ~192 pages pipeline well; on photographs with dark tables in
    // them (the reference's own test images: up to 89 % of a page dark after Otsu) every pass pays its heaviest page - 4.4 s -
    // and two passes cost twice what one would.  A pass is therefore extended for as long as the pages added to it hide behind
    // its heaviest page (cheap census of the dark pixels per page first; constants from profiles/r05: 7.4e6 points 4.4 s alone,
    // 256 pages of 0.84e6 points 1.18 s).  The NL-means balance controller stays off for such a batch.
    std::vector<unsigned> ink;
    if (cp->deskew && env_knobs().chain_overlap && env_knobs().chain_pass == 0 && n_pages > main_sz) {
        constexpr double kTailS = 0.6e-6, kRateS = 5.5e-9;
        st = deskew_ink_census(ctx, n_pages, channels, d_src, src_page_stride, src_step, width, height, &ink, static_cast<hipStream_t>(stream));
        if (st != PRL_OK) return st;
        unsigned heaviest = 0;
        double sum = 0.0;
        int fit = 0;   // pages of the first pass that hide behind the heaviest of them
        for (int i = 0; i < std::min(n_pages, chunk); ++i) {
            heaviest = std::max(heaviest, ink[(size_t)i]);
            sum += ink[(size_t)i];
            if (sum * kRateS <= heaviest * kTailS) fit = i + 1;
        }
        if (fit > main_sz * 5 / 4) {
            main_sz = first_sz = std::min(chunk, fit);
            adaptive = false;
            if (env_knobs().debug)
                std::fprintf(stderr, "[prl chain] tail-bound batch: heaviest page %u points, passes of %d pages\n", heaviest, main_sz);
        }
    }
    // the largest pass the workspace is sized for (an adaptive schedule may grow by a third)
    const int max_cnt = std::max(1, std::min({chunk, n_pages, adaptive ? std::max(main_sz, first_sz) * 4 / 3 : std::max(main_sz, first_sz)}));
    auto next_count = [&](int first) {
        int cnt = std::min({first == 0 ? first_sz : main_sz, n_pages - first, max_cnt});
        if (n_pages - first - cnt > 0 && n_pages - first - cnt < main_sz / 8 && n_pages - first <= max_cnt) cnt = n_pages - first;  // no tiny last pass
        return cnt;
    };
    if (per_page) {
        st = ensure_stage(ctx, per_page * (size_t)max_cnt);
        if (st != PRL_OK) return st;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    st = stage_acquire(ctx, hs);  // another stream's chain / host call may still read the area
    if (st != PRL_OK) return st;
    StageRelease release{ctx, hs};

    // The angle search of deskew (HoughLinesP: one wavefront per page waiting on scattered atomics, seconds per pass) runs
    // for pass k+1 on the side stream, from a helper thread, while the rotation and the other stages of pass k (NL-means:
    // ALU / LDS work that needs no memory bandwidth) run on the caller's stream.  Both only read the source pages.
    struct Finder {
        std::thread th;
        int st = PRL_OK;
        std::string detail;
        DeskewPlan plan;
        SearchStart start;     // lets this thread's caller wait for the Hough kernel of the search to be under way
        double seconds = 0.0;  // how long the search took
        void join() { if (th.joinable()) th.join(); }
        ~Finder() { join(); if (start.ev) (void)hipEventDestroy(start.ev); }
    } finder;
    // Which kernel searches a pass (round 6).  The group kernel (accumulator in LDS) is 3 to 5 times faster than k_ppht_mw but fills the
    // LDS of every CU it runs on: beside this chain's NL-means it does not hide, it takes turns with it.  k_ppht_mw lives on memory-side
    // atomics and does hide behind NL-means - when it is clearly shorter than the body it runs beside.  Measured on 1024 synthetic A4
    // text scans (profiles/r06/chain_1024_schedules.txt): group kernel for every pass, the pass size following the controller down
    // to 64 pages (the tails of search and body interleave) 6.14 s; group kernel, passes of 192-208 pages 6.40 s; k_ppht_mw for the
    // passes whose estimate fits the body (alternating with the group kernel) 6.54 s - its 192-page search takes 1.3 s beside NL-means,
    // not the 0.9 s it takes alone; round 5 (k_ppht_mw throughout) 6.16 s.  So k_ppht_mw is preferred only where its estimate is
    // HALF the body's (pages with few points): costs as measured, k_ppht_mw max(5.5 ns per point of the pass, 0.6 us per point of its
    // heaviest page), NL-means 4.35 ms per 3508 x 3508 x 3 page.
    auto search_prefers_mw = [&](int first, int cnt, int beside_cnt) -> bool {
        if (!cp->denoise || env_knobs().chain_overlap != 2 || beside_cnt <= 0 || ink.empty()) return false;
        double sum = 0.0, heaviest = 0.0;
        for (int i = first; i < first + cnt; ++i) {
            sum += ink[(size_t)i];
            heaviest = std::max(heaviest, (double)ink[(size_t)i]);
        }
        const double est_mw = std::max(sum * 5.5e-9, heaviest * 0.6e-6);
        const double est_nlm = 4.35e-3 * beside_cnt * ((double)len * len) / (3508.0 * 3508.0);
        return est_mw <= 0.5 * est_nlm;
    };
    auto start_find = [&](int first, int cnt, int beside_cnt = 0) {
        finder.st = PRL_OK;
        finder.start.reset();
        finder.start.prefer_mw = search_prefers_mw(first, cnt, beside_cnt);
        if (env_knobs().debug)
            std::fprintf(stderr, "[prl chain] search of pages %d..%d: %s\n", first, first + cnt - 1, finder.start.prefer_mw ? "k_ppht_mw (hides behind NL-means)" : "group kernel");
        finder.th = std::thread([&, first, cnt] {
            struct Done { Finder* f; std::chrono::steady_clock::time_point t; ~Done() { f->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); f->start.finish(); } } done{&finder, std::chrono::steady_clock::now()};
            if (hipSetDevice(dev) != hipSuccess) { finder.st = PRL_ERR_NO_DEVICE; return; }
            const auto t0 = std::chrono::steady_clock::now();
            struct Log { decltype(t0) t; int first; ~Log() { if (env_knobs().debug) std::fprintf(stderr, "[prl chain %.3f] angle search of pages %d..: %.3f s\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(), first, std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); } } log{t0, first};
            finder.st = deskew_find(ctx, cnt, channels, d_src + (size_t)first * src_page_stride, src_page_stride, src_step, width,
                                    height, &finder.plan, ctx->side, &finder.start);
            if (finder.st != PRL_OK) finder.detail = prl_hip_last_error_detail();
        });
    };
    if (cp->deskew) {
        if (!ctx->side) {
            PRL_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->side_ev, hipEventDisableTiming));
        }
        PRL_HIP_CHECK(hipEventRecord(ctx->side_ev, hs));  // the source pages may come from earlier work on the caller's stream
        PRL_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
        PRL_HIP_CHECK(hipEventCreateWithFlags(&finder.start.ev, hipEventDisableTiming));
    }
    // Beside a search only runs what is enqueued through this: the caller's stream waits (on the device) for the streaming prelude
    // of the search to be through, the host for its Hough kernel to be submitted - a search that starts next to other kernels is
    // slowed for good, one whose wavefronts are resident first runs at its own speed.  No wall-clock guesses.
    auto behind_search_start = [&]() -> int {
        if (finder.start.wait()) PRL_HIP_CHECK(hipStreamWaitEvent(hs, finder.start.ev, 0));
        return PRL_OK;
    };
    struct EventOwner { hipEvent_t e = nullptr; ~EventOwner() { if (e) (void)hipEventDestroy(e); } } body_done;
    if (cp->deskew && cp->denoise && env_knobs().chain_overlap == 2) {
        // the Lab planes of the largest pass, up front: growing the buffer later synchronises the device under a running search
        st = ensure_buffer(&ctx->chain_planes, &ctx->chain_planes_bytes, denoise_plane_bytes(len, len) * (size_t)max_cnt);
        if (st != PRL_OK) return st;
    }
    std::vector<int32_t> wh((size_t)max_cnt * 2);
    DeskewPlan plan;
    int first = 0, cnt = next_count(0);
    if (cp->deskew) start_find(0, cnt);
    for (; first < n_pages;) {
        const int nfirst = first + cnt;
        int ncnt = 0;   // size of the next pass: fixed when its search starts
        uint8_t* ws = static_cast<uint8_t*>(ctx->stage);
        const uint8_t* cur = d_src + (size_t)first * src_page_stride;
        size_t cur_ps = src_page_stride, cur_step = src_step;
        const auto t_pass = std::chrono::steady_clock::now();
        if (cp->deskew) {
            finder.join();
            if (env_knobs().debug)
                std::fprintf(stderr, "[prl chain %.3f] pass at page %d waited %.3f s for its angles\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(), first,
                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pass).count());
            if (finder.st != PRL_OK) {
                set_error_detail(finder.detail);
                return finder.st;
            }
            plan = std::move(finder.plan);
            finder.plan = DeskewPlan();
            const bool overlap = env_knobs().chain_overlap != 0;
            const bool later = env_knobs().chain_overlap == 2 && cp->denoise;   // split mode starts it after the streaming head of the pass
            if (overlap && !later && nfirst < n_pages) {
                ncnt = next_count(nfirst);
                start_find(nfirst, ncnt);
                st = behind_search_start();   // this pass's own work starts once the search is under way
                if (st != PRL_OK) return st;
            }
            uint8_t* desk = ws;
            ws = desk + desk_page * (size_t)cnt;
            st = deskew_apply(ctx, plan, cnt, channels, cur, cur_ps, cur_step, width, height, desk, desk_page, (size_t)len * channels, hs);
            if (st != PRL_OK) return st;
            std::copy(plan.wh.begin(), plan.wh.end(), wh.begin());
            if (angles) std::copy(plan.angles.begin(), plan.angles.end(), angles + first);
            cur = desk; cur_ps = desk_page; cur_step = (size_t)len * channels;
        } else {
            for (int i = 0; i < cnt; ++i) { wh[2 * (size_t)i] = width; wh[2 * (size_t)i + 1] = height; }
            if (angles) for (int i = 0; i < cnt; ++i) angles[first + i] = 0.0;
        }
        // runs of equal page size, each with its own slice of the workspace
        struct Run { int r0, r1, pw, ph; RunState rs; uint8_t* ws; uint8_t* planes; };
        std::vector<Run> runs;
        // chain_overlap == 2 (with deskew and denoise): a pass is cut in three.  HEAD - rotation and BGR->Lab, streaming kernels -
        // runs before the search of the next pass is started; BODY - the NL-means kernels, compute - beside that search;
        // TAIL - Lab->BGR and the stages after denoise, streaming again - after it.  Streaming kernels get a fraction of their
        // bandwidth while a search's atomics are in flight (16-50x their time in its first second) and hold up whatever is
        // queued behind them, and a search that starts beside them is slowed for good.
        const bool split = cp->deskew && cp->denoise && env_knobs().chain_overlap == 2;
        size_t planes_need = 0;
        {
            uint8_t* wsr = ws;
            for (int r0 = 0; r0 < cnt;) {
                int r1 = r0 + 1;
                while (r1 < cnt && wh[2 * (size_t)r1] == wh[2 * (size_t)r0] && wh[2 * (size_t)r1 + 1] == wh[2 * (size_t)r0 + 1]) ++r1;
                Run run{r0, r1, wh[2 * (size_t)r0], wh[2 * (size_t)r0 + 1], {}, wsr, nullptr};
                prl_binarize_geometry g;
                st = prl_hip_binarize_geometry(&cp->binarize, run.pw, run.ph, &g);
                if (st != PRL_OK) return st;
                wsr += chain_layout(cp, channels, run.pw, run.ph, g).total * (size_t)(r1 - r0);
                planes_need += denoise_plane_bytes(run.pw, run.ph) * (size_t)(r1 - r0);
                for (int i = r0; i < r1; ++i) {
                    out_wh[2 * (size_t)(first + i)] = g.out_w;
                    out_wh[2 * (size_t)(first + i) + 1] = g.out_h;
                }
                runs.push_back(run);
                r0 = r1;
            }
        }
        if (!split) {
            for (Run& run : runs) {
                st = chain_uniform_a(cp, run.r1 - run.r0, channels, cur + (size_t)run.r0 * cur_ps, cur_ps, cur_step, run.pw, run.ph, run.ws, stream,
                                     &run.rs);
                if (st == PRL_OK)
                    st = chain_uniform_b(cp, run.r1 - run.r0, channels, run.rs, run.pw, run.ph, d_dst + (size_t)(first + run.r0) * dst_page_stride,
                                         dst_page_stride, dst_step, stream);
                if (st != PRL_OK) return st;
            }
        } else {
            st = ensure_buffer(&ctx->chain_planes, &ctx->chain_planes_bytes, planes_need);
            if (st != PRL_OK) return st;
            uint8_t* pl = static_cast<uint8_t*>(ctx->chain_planes);
            for (Run& run : runs) {   // HEAD (the rotation was enqueued above)
                run.planes = pl;
                pl += denoise_plane_bytes(run.pw, run.ph) * (size_t)(run.r1 - run.r0);
                st = denoise_convert_in(ctx, run.r1 - run.r0, channels, cur + (size_t)run.r0 * cur_ps, cur_ps, cur_step, run.pw, run.ph, run.planes, hs);
                if (st != PRL_OK) return st;
            }
            const bool beside = nfirst < n_pages;
            if (beside) {
                PRL_HIP_CHECK(hipStreamSynchronize(hs));   // the head is through before the next search takes the memory system
                ncnt = next_count(nfirst);
                start_find(nfirst, ncnt, cnt);
                st = behind_search_start();
                if (st != PRL_OK) return st;
            }
            for (Run& run : runs) {   // BODY
                st = denoise_nlm(ctx, run.r1 - run.r0, cp->denoise_strength, run.planes, run.pw, run.ph, hs);
                if (st != PRL_OK) return st;
            }
            if (beside) {
                if (!body_done.e) PRL_HIP_CHECK(hipEventCreateWithFlags(&body_done.e, hipEventDisableTiming));
                PRL_HIP_CHECK(hipEventRecord(body_done.e, hs));
            }
            finder.join();
            if (beside) {
                // how the body compared with the search beside it sizes the searches that are still to start
                const bool early = hipEventQuery(body_done.e) == hipSuccess;
                if (!early) (void)hipGetLastError();   // ("not ready" is an answer, not an error to find later)
                const auto tj = std::chrono::steady_clock::now();
                if (!early) PRL_HIP_CHECK(hipEventSynchronize(body_done.e));   // (the tail is enqueued behind the body anyway)
                const double past = early ? 0.0 : std::chrono::duration<double>(std::chrono::steady_clock::now() - tj).count();
                if (adaptive && finder.seconds > 0.0 && ncnt >= main_sz) {   // (a short last search says nothing about the balance)
                    const int before = main_sz;
                    if (past > 0.03 * finder.seconds) main_sz = (int)(main_sz * finder.seconds / (finder.seconds + past)) / 16 * 16;
                    else if (early) main_sz += 16;
                    main_sz = std::max(std::min(64, max_cnt), std::min(main_sz, max_cnt));
                    if (env_knobs().debug && main_sz != before)
                        std::fprintf(stderr, "[prl chain] pass size %d -> %d (search %.3f s, NL-means %.3f s past it)\n", before, main_sz, finder.seconds, past);
                }
                if (env_knobs().debug)
                    std::fprintf(stderr, "[prl chain] pass at page %d: NL-means ran %.3f s past the search\n", first, past);
            }
            for (Run& run : runs) {   // TAIL
                prl_binarize_geometry g;
                st = prl_hip_binarize_geometry(&cp->binarize, run.pw, run.ph, &g);
                if (st != PRL_OK) return st;
                const ChainLayout l = chain_layout(cp, channels, run.pw, run.ph, g);
                const int rc = run.r1 - run.r0;
                st = denoise_convert_out(ctx, rc, channels, run.planes, run.pw, run.ph, run.ws, l.denoised, (size_t)run.pw * channels, hs);
                if (st != PRL_OK) return st;
                run.rs.cur = run.ws; run.rs.cur_ps = l.denoised; run.rs.cur_step = (size_t)run.pw * channels;
                run.rs.w = run.ws + l.denoised * (size_t)rc;
                st = chain_uniform_b(cp, rc, channels, run.rs, run.pw, run.ph, d_dst + (size_t)(first + run.r0) * dst_page_stride, dst_page_stride,
                                     dst_step, stream);
                if (st != PRL_OK) return st;
            }
        }
        if (cp->deskew && !env_knobs().chain_overlap && nfirst < n_pages) {
            PRL_HIP_CHECK(hipStreamSynchronize(hs));
            ncnt = next_count(nfirst);
            start_find(nfirst, ncnt);
            finder.join();
        }
        if (env_knobs().debug) {
            (void)hipStreamSynchronize(hs);
            std::fprintf(stderr, "[prl chain %.3f] pass at page %d (%d pages) done after %.3f s\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(), first, cnt,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pass).count());
        }
        first = nfirst;
        cnt = ncnt > 0 ? ncnt : (first < n_pages ? next_count(first) : 0);
    }
    return PRL_OK;
}

int prl_hip_chain_max_out_size(const prl_chain_params* cp, int width, int height, int* out_w, int* out_h)
{
    if (!cp || !out_w || !out_h) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    const int len = std::max(width, height);
    prl_binarize_geometry g;
    int st = prl_hip_binarize_geometry(&cp->binarize, cp->deskew ? len : width, cp->deskew ? len : height, &g);
    if (st != PRL_OK) return st;
    *out_w = g.out_w;
    *out_h = g.out_h;
    if (cp->deskew) {  // pages without an angle keep width x height: their result can be larger in one dimension only if ... never
        prl_binarize_geometry g2;
        st = prl_hip_binarize_geometry(&cp->binarize, width, height, &g2);
        if (st != PRL_OK) return st;  // (a page that keeps its size must be binarizable too)
        *out_w = std::max(*out_w, g2.out_w);
        *out_h = std::max(*out_h, g2.out_h);
    }
    return PRL_OK;
}

// Uniform result size: the chain without deskew.  d_dst receives out_w x out_h bytes per page (prl_hip_binarize_geometry).
int prl_hip_chain_batch_device(const prl_chain_params* cp, int n_pages, int channels, const uint8_t* d_src,
                               size_t src_page_stride, size_t src_step, int width, int height, uint8_t* d_dst,
                               size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (cp && cp->deskew) return PRL_ERR_BAD_ARG;  // per-page result sizes: prl_hip_chain_pages_device
    int st = chain_check(cp, n_pages, channels, d_src, src_step, width, height, d_dst);
    if (st != PRL_OK) return st;
    std::vector<int32_t> wh((size_t)std::max(n_pages, 1) * 2);
    return prl_hip_chain_pages_device(cp, n_pages, channels, d_src, src_page_stride, src_step, width, height, d_dst, dst_page_stride,
                                      dst_step, wh.data(), nullptr, stream);
}

void prl_hip_default_chain_params(prl_chain_params* out)
{
    if (!out) return;
    out->denoise = 0;
    out->denoise_strength = 5.5f;  // src/denoise/denoiseNLM.h:32
    prl_hip_default_params(PRL_SAUVOLA, &out->binarize);
    out->thin = PRL_CHAIN_NO_THINNING;
    out->deskew = 0;
    out->background_normalization = 0;
}

}  // extern "C"
