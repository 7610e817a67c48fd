// thin.hip — prl::thinZhangSuen (src/thinning/thinZhangSuen.cpp:15-108) and prl::thinGuoHall
// (src/thinning/thinGuoHall.cpp:15-107): SURVEY.md §8f rank 1, the last step of the BASELINE config-5 chain.
//
// The reference works on a 0/1 byte image (`&= 1`, :83) and repeats two sub-iterations until a whole pass
// changes nothing (:88-96).  Both sub-iterations are pure 3x3 boolean functions of the neighbourhood, so the
// device keeps the image as ONE BIT per pixel (a 4K page is 2 MiB and stays in L2 across passes) and evaluates
// 32 pixels per thread with bit-sliced logic:
//   neighbours p2..p9 (thinZhangSuen.cpp:28-35) are the three row words shifted by -1/0/+1 bit;
//   Zhang-Suen: A == 1 (exactly one 0->1 transition around the ring) via an "any / two-or-more" pair,
//               2 <= B <= 6 via a bit-sliced population count, m1 == 0, m2 == 0      (:37-51)
//   Guo-Hall:   C == 1, 2 <= min(N1, N2) <= 3, m == 0                                 (thinGuoHall.cpp:40-50)
//   marker applied as `image &= ~marker` after the sub-iteration (:54) = ping-pong between two bit planes;
//   border pixels (row/column 0 and last) are never marked (:22-24).
// Integer/boolean throughout: bit-exact by construction.  Bound: latency/launch (each sub-iteration touches
// 1/8 B per pixel out of L2); HBM traffic is one u8 read (pack) and one u8 write (unpack) per pixel.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {

namespace {

// pack: bit = pixel & 1  (imageUnderProcess &= 1, thinZhangSuen.cpp:83).  One thread per 32-pixel word.  Rows of a
// (W-1)-wide binarizer output are rarely 8-byte aligned, so the thread reads the (up to five) ALIGNED u64 that cover
// its 32 bytes and funnel-shifts them; an aligned u64 that holds at least one byte of the row never leaves the
// row's memory page, so the over-read is safe, and bytes past the row end are masked off.
// flip = 0xffffffff: pack cv::bitwise_not of the source (the chain feeds the binarizer's mask, white = background,
// into thinning, which thins white: binarizeSauvola -> bitwise_not -> thinZhangSuen without a separate inversion pass)
__global__ void __launch_bounds__(256) k_thin_pack(PageSet src, int width, int height, int wpr, unsigned* __restrict__ bits,
                                                  size_t plane_words, unsigned flip)
{
    const int page = blockIdx.y;
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= plane_words) return;
    const int y = (int)(gid / (unsigned)wpr), k = (int)(gid - (unsigned)y * (unsigned)wpr);
    const uint8_t* p0 = src.page(page) + (size_t)y * src.step + (size_t)k * 32;
    const int nb = min(32, width - k * 32);           // valid bytes of this word (>= 1)
    const int a = (int)((size_t)p0 & 7);
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p0 - a);
    unsigned long long v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = (8 * i - a < nb) ? q[i] : 0ull;   // loaded only if it holds a valid byte
    const int sh = 8 * a;
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned long long u = sh ? ((v[i] >> sh) | (v[i + 1] << (64 - sh))) : v[i];
        // bit 0 of each of the 8 bytes -> 8 adjacent bits (byte j lands on bit j)
        w |= (unsigned)(((u & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * i);
    }
    w ^= flip;
    if (nb < 32) w &= (1u << nb) - 1u;
    bits[(size_t)page * plane_words + gid] = w;
}

// unpack: byte = bit * 255  (outputImage = imageUnderProcess * 255, :100-106).  One thread per ALIGNED 32-byte block
// of the destination row (wpr + 1 blocks per row: the first one may start up to 7 pixels left of the row); its 32
// bits straddle two plane words.  Whole u64 stores inside the row, byte stores on the two ragged ends.
__global__ void __launch_bounds__(256) k_thin_unpack(const unsigned* __restrict__ bits, size_t plane_words, int wpr,
                                                    PageSetOut dst, int width, int height)
{
    const int page = blockIdx.y;
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned bpr = (unsigned)wpr + 1u;
    if (gid >= bpr * (unsigned)height) return;
    const int y = (int)(gid / bpr), t = (int)(gid - (unsigned)y * bpr);
    uint8_t* row = dst.page(page) + (size_t)y * dst.step;
    const int a = (int)((size_t)row & 7);
    const int xs = 32 * t - a;                        // first pixel of this block
    const unsigned* wrow = bits + (size_t)page * plane_words + (size_t)y * wpr;
    const unsigned lo = (t >= 1) ? wrow[t - 1] : 0u, hi = (t < wpr) ? wrow[t] : 0u;
    const unsigned w = a ? ((lo >> (32 - a)) | (hi << a)) : hi;   // a == 0: block t is word t
    if (xs >= width) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // 8 bits -> 8 bytes of 0x00 / 0xFF (bit j to byte j)
        const unsigned long long b8 = (w >> (8 * i)) & 0xffull;
        const unsigned long long sel = (b8 * 0x0101010101010101ull) & 0x8040201008040201ull;  // byte j keeps bit j
        const unsigned long long nz = (sel + 0x7f7f7f7f7f7f7f7full) & 0x8080808080808080ull;  // byte non-zero -> bit 7
        const unsigned long long o = (nz >> 7) * 255ull;
        const int x = xs + 8 * i;
        if (x >= 0 && x + 8 <= width) {
            *reinterpret_cast<unsigned long long*>(row + x) = o;
        } else {
            for (int j = 0; j < 8; ++j)
                if (x + j >= 0 && x + j < width) row[x + j] = (uint8_t)(o >> (8 * j));
        }
    }
}

// exactly one of the given bit-planes set, per bit
struct OneOf {
    unsigned any = 0, two = 0;
    __device__ __forceinline__ void add(unsigned t)
    {
        two |= any & t;
        any |= t;
    }
    __device__ __forceinline__ unsigned exactly_one() const { return any & ~two; }
};

// bit-sliced counter (values 0..8 in s3 s2 s1 s0) of up to 8 one-bit planes
struct Count8 {
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    __device__ __forceinline__ void add(unsigned t)
    {
        const unsigned c0 = s0 & t;
        s0 ^= t;
        const unsigned c1 = s1 & c0;
        s1 ^= c0;
        const unsigned c2 = s2 & c1;
        s2 ^= c1;
        s3 |= c2;
    }
};

// Marker bits of one sub-iteration for the 32 pixels of word `c`, given the eight surrounding words.
// METHOD 0 = Zhang-Suen, 1 = Guo-Hall; `iteration` as in the reference.
template <int METHOD>
__device__ __forceinline__ unsigned thin_mark(int iteration, unsigned c, unsigned up, unsigned dn, unsigned cl, unsigned cr,
                                              unsigned ul, unsigned ur, unsigned dl, unsigned dr)
{
    // bit b of each plane = neighbour of pixel x = 32k + b      (thinZhangSuen.cpp:28-35)
    const unsigned p2 = up;                                  // (i-1, j)
    const unsigned p3 = (up >> 1) | (ur << 31);              // (i-1, j+1)
    const unsigned p4 = (c >> 1) | (cr << 31);               // (i,   j+1)
    const unsigned p5 = (dn >> 1) | (dr << 31);              // (i+1, j+1)
    const unsigned p6 = dn;                                  // (i+1, j)
    const unsigned p7 = (dn << 1) | (dl >> 31);              // (i+1, j-1)
    const unsigned p8 = (c << 1) | (cl >> 31);               // (i,   j-1)
    const unsigned p9 = (up << 1) | (ul >> 31);              // (i-1, j-1)
    if (METHOD == 0) {
        OneOf A;                                             // :37-40
        A.add(~p2 & p3); A.add(~p3 & p4); A.add(~p4 & p5); A.add(~p5 & p6);
        A.add(~p6 & p7); A.add(~p7 & p8); A.add(~p8 & p9); A.add(~p9 & p2);
        Count8 B;                                            // :42
        B.add(p2); B.add(p3); B.add(p4); B.add(p5); B.add(p6); B.add(p7); B.add(p8); B.add(p9);
        const unsigned ge2 = B.s3 | B.s2 | B.s1;
        const unsigned le6 = ~(B.s3 | (B.s2 & B.s1 & B.s0));
        const unsigned m1 = iteration == 0 ? (p2 & p4 & p6) : (p2 & p4 & p8);   // :44
        const unsigned m2 = iteration == 0 ? (p4 & p6 & p8) : (p2 & p6 & p8);   // :45
        return A.exactly_one() & ge2 & le6 & ~m1 & ~m2;      // :47
    } else {
        OneOf Cn;                                            // thinGuoHall.cpp:40-41
        Cn.add(~p2 & (p3 | p4)); Cn.add(~p4 & (p5 | p6)); Cn.add(~p6 & (p7 | p8)); Cn.add(~p8 & (p9 | p2));
        Count8 N1, N2;                                       // :42-43
        N1.add(p9 | p2); N1.add(p3 | p4); N1.add(p5 | p6); N1.add(p7 | p8);
        N2.add(p2 | p3); N2.add(p4 | p5); N2.add(p6 | p7); N2.add(p8 | p9);
        const unsigned n1_ge2 = N1.s2 | N1.s1, n2_ge2 = N2.s2 | N2.s1;
        const unsigned n1_le3 = ~N1.s2, n2_le3 = ~N2.s2;
        const unsigned n_ok = n1_ge2 & n2_ge2 & (n1_le3 | n2_le3);              // 2 <= min(N1,N2) <= 3   :44,47
        const unsigned m = iteration == 0 ? ((p6 | p7 | ~p9) & p8) : ((p2 | p3 | ~p5) & p4);   // :45
        return Cn.exactly_one() & n_ok & ~m;
    }
}

// One whole pass (both sub-iterations, thinZhangSuen.cpp:90-91) in one launch.  A wavefront streams down a strip of 64
// words: a lane keeps the last three input rows of its word, gets the left / right words from its neighbour lanes
// (DPP wave_shr/shl:1), forms sub-iteration 1 for the middle row, keeps the last three such rows and forms
// sub-iteration 2 one row behind - registers only, one load and one store per word and pass (a kernel per
// sub-iteration reads nine words per word and needs two launches).  Lanes 0,1 / 62,63 and two rows above / below a segment are
// halo.
constexpr int kThinUseful = 60;  // output words per 64-lane strip (lanes 2 .. 61)

__device__ __forceinline__ unsigned lane_dn(unsigned v)  // word of lane - 1 (0 at the wavefront edge)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned lane_up(unsigned v)  // word of lane + 1
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

template <int METHOD>
__global__ void __launch_bounds__(256) k_thin_pass(const unsigned* __restrict__ in, unsigned* __restrict__ out,
                                                  size_t plane_words, int wpr, int width, int height, int n_strips,
                                                  int n_segs, int rows_per_seg, unsigned total_waves,
                                                  unsigned* __restrict__ changed, const unsigned* __restrict__ done,
                                                  const unsigned char* __restrict__ act_prev,
                                                  unsigned char* __restrict__ act_cur)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wv);
    if (wid >= total_waves) return;
    const int per_page = n_strips * n_segs;
    const int page = (int)(wid / (unsigned)per_page);
    if (done[page]) return;  // converged: its bit plane is final and stays in buffer A
    const int rem = (int)(wid - (unsigned)page * (unsigned)per_page);
    const int seg = rem / n_strips, strip = rem - seg * n_strips;
    // Activity tracking: a tile (strip x segment) whose own and eight neighbouring tiles came out of the previous pass
    // unchanged sees the same input (tile + 2-row / 2-word halo) as that pass did, so it would reproduce its input -
    // which both ping-pong buffers already hold.  Such tiles are skipped; after the first few passes that is most
    // of the page (only the thickest strokes keep shrinking).
    {
        bool active = false;
        for (int ds = -1; ds <= 1; ++ds)
            for (int dt = -1; dt <= 1; ++dt) {
                const int s2 = seg + ds, t2 = strip + dt;
                if (s2 >= 0 && s2 < n_segs && t2 >= 0 && t2 < n_strips)
                    active |= act_prev[(size_t)page * per_page + (size_t)s2 * n_strips + t2] != 0;
            }
        if (!active) {
            if (lane == 0) act_cur[wid] = 0;
            return;
        }
    }
    const int k = strip * kThinUseful - 2 + lane;  // this lane's word of the row
    const bool kin = k >= 0 && k < wpr;
    const unsigned* pin = in + (size_t)page * plane_words + (kin ? k : 0);
    unsigned* pout = out + (size_t)page * plane_words + (kin ? k : 0);
    const int ys = seg * rows_per_seg, ye = min(ys + rows_per_seg, height);
    // columns 0 and width-1 are never marked; bits beyond the row are zero anyway
    unsigned col_ok = 0xffffffffu;
    if (k == 0) col_ok &= ~1u;
    if (((width - 1) >> 5) == k) col_ok &= ~(1u << ((width - 1) & 31));

    auto fetch = [&](int r) -> unsigned { return (kin && r >= 0 && r < height) ? pin[(size_t)r * wpr] : 0u; };

    // input rows r-2, r-1, r with their neighbour words; sub-iteration-1 rows r-3, r-2, r-1 likewise
    unsigned a0 = 0, a0l = 0, a0r = 0, a1 = 0, a1l = 0, a1r = 0;          // a0 = in[r-2], a1 = in[r-1]
    unsigned b0 = 0, b0l = 0, b0r = 0, b1 = 0, b1l = 0, b1r = 0;          // b0 = I1[r-3], b1 = I1[r-2]
    bool any_change = false;
    unsigned vnext = fetch(ys - 2);
#pragma unroll 1
    for (int r = ys - 2; r <= ye + 1; ++r) {
        const unsigned a2 = vnext;
        vnext = fetch(r + 1);
        const unsigned a2l = lane_dn(a2), a2r = lane_up(a2);
        // sub-iteration 1 for row r-1 (rows 0 and height-1 are never marked)
        // (a word without foreground cannot lose a pixel: whole wavefronts of background skip the logic)
        unsigned i1 = a1;
        if (r - 1 >= 1 && r - 1 <= height - 2 && __ballot(a1 != 0u) != 0ull)
            i1 = a1 & ~(thin_mark<METHOD>(0, a1, a0, a2, a1l, a1r, a0l, a0r, a2l, a2r) & col_ok);
        const unsigned i1l = lane_dn(i1), i1r = lane_up(i1);
        // sub-iteration 2 for row r-2
        unsigned o = b1;
        if (r - 2 >= 1 && r - 2 <= height - 2 && __ballot(b1 != 0u) != 0ull)
            o = b1 & ~(thin_mark<METHOD>(1, b1, b0, i1, b1l, b1r, b0l, b0r, i1l, i1r) & col_ok);
        const int ro = r - 2;
        if (ro >= ys && ro < ye && lane >= 2 && lane <= 61 && kin) {
            pout[(size_t)ro * wpr] = o;
            any_change |= (o != a0);  // a0 = in[r-2]: the word this pass started from
        }
        // shift the windows
        a0 = a1; a0l = a1l; a0r = a1r;
        a1 = a2; a1l = a2l; a1r = a2r;
        b0 = b1; b0l = b1l; b0r = b1r;
        b1 = i1; b1l = i1l; b1r = i1r;
    }
    const bool tile_changed = __ballot(any_change) != 0ull;
    if (lane == 0) {
        act_cur[wid] = tile_changed ? 1 : 0;
        if (tile_changed) changed[page] = 1u;
    }
}

// after both sub-iterations of a pass: a page whose pass changed nothing is final (do-while test, :93-96)
__global__ void k_thin_endpass(unsigned* changed, unsigned* done, int n_pages)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pages) {
        if (changed[i] == 0) done[i] = 1;
        changed[i] = 0;
    }
}

}  // namespace
}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

int prl_hip_thin_batch_device(int method, int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                              int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    return prl_hip::thin_batch_device(method, n_pages, d_src, src_page_stride, src_step, width, height, d_dst, dst_page_stride,
                                      dst_step, stream, false);
}

}  // extern "C"

namespace prl_hip {
int thin_batch_device(int method, int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step, int width,
                      int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream, bool invert_input)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;  // "Input image for thinning is empty" (thinZhangSuen.cpp:59-62)
    if (method != PRL_THIN_ZHANGSUEN && method != PRL_THIN_GUOHALL) return PRL_ERR_BAD_ARG;
    if (n_pages < 0 || !d_src || !d_dst || src_step < (size_t)width || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if (n_pages > 32768) {  // the page index sits in a grid dimension limited to 65535
        for (int first = 0; first < n_pages; first += 32768) {
            const int st2 = thin_batch_device(method, std::min(32768, n_pages - first),
                                              d_src + (size_t)first * src_page_stride, src_page_stride, src_step, width,
                                              height, d_dst + (size_t)first * dst_page_stride, dst_page_stride, dst_step,
                                              stream, invert_input);
            if (st2 != PRL_OK) return st2;
        }
        return PRL_OK;
    }
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);

    const int wpr = (width + 31) / 32;
    const size_t plane_words = (size_t)wpr * height;
    if (plane_words >= 0x7fffff00ull) return PRL_ERR_BAD_ARG;  // kernels index a page's words with 32 bits
    const size_t bits_bytes = plane_words * sizeof(unsigned) * (size_t)n_pages;
    const size_t flags_bytes = ((size_t)n_pages * sizeof(unsigned) + 255) / 256 * 256;
    // one launch per pass: strips of 60 words x segments of rows, one wavefront each
    const int n_strips = (wpr + kThinUseful - 1) / kThinUseful;
    int rps = 256;  // measured (16 A4 pages): 32 rows 31 us per pass, 16: 26, 8: 26, 4: 31
    while (rps > 16 && (long long)n_pages * n_strips * ((height + rps - 1) / rps) < 16384) rps /= 2;
    if (env_knobs().thin_rps) rps = env_knobs().thin_rps;
    const int n_segs = (height + rps - 1) / rps;
    const unsigned long long tw = (unsigned long long)n_pages * n_strips * n_segs;
    if (tw >= 0xfffffff0ull) return PRL_ERR_BAD_ARG;
    const size_t act_cap = ((size_t)tw + 255) / 256 * 256;  // one activity byte per tile and pass parity
    st = ensure_scratch(ctx, 2 * bits_bytes + 2 * flags_bytes + 2 * act_cap);
    if (st != PRL_OK) return st;
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(s, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    auto* A = static_cast<unsigned*>(ctx->scratch);
    auto* B = A + plane_words * (size_t)n_pages;
    auto* changed = reinterpret_cast<unsigned*>(reinterpret_cast<uint8_t*>(B) + bits_bytes);
    auto* done = reinterpret_cast<unsigned*>(reinterpret_cast<uint8_t*>(changed) + flags_bytes);
    PRL_HIP_CHECK(hipMemsetAsync(changed, 0, 2 * flags_bytes, s));
    auto* act0 = reinterpret_cast<unsigned char*>(done) + flags_bytes;
    auto* act1 = act0 + act_cap;
    PRL_HIP_CHECK(hipMemsetAsync(act0, 1, act_cap, s));  // before the first pass every tile counts as changed

    PageSet ps{};
    ps.base = d_src;
    ps.page_stride = src_page_stride;
    ps.step = src_step;
    PageSetOut pd{};
    pd.base = d_dst;
    pd.page_stride = dst_page_stride;
    pd.step = dst_step;
    const dim3 gw((unsigned)((plane_words + 255) / 256), n_pages);  // one thread per 32-pixel word
    hipLaunchKernelGGL(k_thin_pack, gw, dim3(256), 0, s, ps, width, height, wpr, A, plane_words, invert_input ? 0xffffffffu : 0u);
    PRL_HIP_CHECK(hipGetLastError());

    st = ensure_pinned(ctx, sizeof(unsigned) * (size_t)n_pages);  // flag readback through pinned memory
    if (st != PRL_OK) return st;
    unsigned* h_done = static_cast<unsigned*>(ctx->pinned);
    // No cap on the number of passes: the reference loops until a pass changes nothing (thinZhangSuen.cpp:93-104), and a
    // 2-pixel diagonal stroke erodes from its ends only, about a row per pass (max(W,H)+2 was an unproven bound
    // that returned a non-converged plane).  Every effective pass removes at least one pixel, so the loop ends.
    int group = 4;  // passes per host check, doubled each time (a pass over converged pages / idle tiles is nearly free)
    const unsigned wpb = (unsigned)env_knobs().thin_wpb;
    const dim3 gp((unsigned)((tw + wpb - 1) / wpb)), bp(64 * wpb);  // short wavefronts: 4 per workgroup measured best (26.1 vs 28.4 us)
    // Passes alternate A -> B -> A.  The pass that finds a page unchanged has just written a copy of its input, so
    // from then on BOTH buffers hold that page's final plane (later passes skip it): k_thin_unpack can always read A.
    for (int pass = 0;;) {
        for (int gidx = 0; gidx < group; ++gidx, ++pass) {
            const unsigned* in = (pass & 1) ? B : A;
            unsigned* out = (pass & 1) ? A : B;
            const unsigned char* act_prev = (pass & 1) ? act1 : act0;
            unsigned char* act_cur = (pass & 1) ? act0 : act1;
            if (method == PRL_THIN_ZHANGSUEN)
                hipLaunchKernelGGL(k_thin_pass<0>, gp, bp, 0, s, in, out, plane_words, wpr, width, height, n_strips,
                                   n_segs, rps, (unsigned)tw, changed, done, act_prev, act_cur);
            else
                hipLaunchKernelGGL(k_thin_pass<1>, gp, bp, 0, s, in, out, plane_words, wpr, width, height, n_strips,
                                   n_segs, rps, (unsigned)tw, changed, done, act_prev, act_cur);
            PRL_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(k_thin_endpass, dim3((n_pages + 255) / 256), dim3(256), 0, s, changed, done, n_pages);
            PRL_HIP_CHECK(hipGetLastError());
        }
        PRL_HIP_CHECK(hipMemcpyAsync(h_done, done, sizeof(unsigned) * (size_t)n_pages, hipMemcpyDeviceToHost, s));
        PRL_HIP_CHECK(hipStreamSynchronize(s));
        if (std::all_of(h_done, h_done + n_pages, [](unsigned v) { return v != 0; })) break;
        group = std::min(group * 2, 32);
    }
    const dim3 gu((unsigned)(((size_t)(wpr + 1) * height + 255) / 256), n_pages);  // one thread per aligned 32-byte block
    hipLaunchKernelGGL(k_thin_unpack, gu, dim3(256), 0, s, A, plane_words, wpr, pd, width, height);
    PRL_HIP_CHECK(hipGetLastError());
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, s));
    return PRL_OK;
}
}  // namespace prl_hip

extern "C" {

int prl_hip_thin_host(int method, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (!src || !dst || src_step < (size_t)width || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    const size_t bytes = ((size_t)width * height + 255) / 256 * 256;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> slk(ctx->stage_mu);  // cached device + pinned staging (lock order: stage_mu, then mu)
    st = ensure_stage(ctx, bytes);
    if (st != PRL_OK) return st;
    st = ensure_stage_pinned(ctx, bytes);
    if (st != PRL_OK) return st;
    uint8_t* d = static_cast<uint8_t*>(ctx->stage);
    DrainOnExit drain_guard{nullptr};   // (direct DMA from the caller's pinned page: see prl_internal.h)
    st = stage_upload(ctx, 0, src, src_step, (size_t)width, height, d, nullptr);
    if (st != PRL_OK) return st;
    st = prl_hip_thin_batch_device(method, 1, d, bytes, (size_t)width, width, height, d, bytes, (size_t)width, nullptr);
    if (st != PRL_OK) return st;
    return stage_download(ctx, 0, d, (size_t)width, height, dst, dst_step, nullptr);
}

}  // extern "C"
