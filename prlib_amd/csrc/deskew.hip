// deskew.hip — prl::deskew and prl::rotate for pages resident in device memory.
//
// Reference: src/deskew/deskew.cpp:208-251 (gray -> Otsu -> findAngle -> rotate), :139-205 (findAngle = bitwise_not,
// cv::HoughLinesP(1, CV_PI/180, 100, width/8.f, 20), angle vote), src/rotate.cpp:35-72.  OpenCV's arithmetic
// [upstream] is restated with citations in the test oracle (oracle/, deskew file); the kernels below reproduce it bit for bit:
//
//   k_hist / k_otsu      256-bin histogram (LDS-privatised) and getThreshVal_Otsu_8u's float64 scan, one thread per page
//   k_dark_mask          mask = (p <= thr) (the bitwise_not of the thresholded page, as 0/1 bytes = HoughLinesP's own
//                        `mask` matrix) + the number of set pixels per row
//   k_row_offsets        exclusive scan of the row counts (one workgroup per page) -> where each row's points start
//   k_collect            HoughLinesP stage 1: the non-zero points in raster order, packed x | y << 16
//   k_ppht               HoughLinesP stage 2, ONE WAVEFRONT PER PAGE.  The progressive probabilistic Hough transform is
//                        sequential by construction (every random point votes into the accumulator the previous points
//                        left, and a detected line erases points and takes their votes back), so a page cannot be split;
//                        the batch dimension supplies the parallelism (1024 pages = one wavefront per SIMD).  Inside a
//                        page the 180 angles of a vote are spread over the lanes (3 per lane, `global_atomic_add` with
//                        return, owner-lane rows), the maximum is a wavefront reduction, and the two line walks test 64
//                        steps per memory round trip (ballot + scalar run-length scan) instead of one.  cv::RNG's
//                        multiply-with-carry sequence is reproduced exactly, so the segments - and therefore the angle -
//                        are those OpenCV finds.  Latency-bound by design: ~3 dependent L2/HBM round trips per point.
//   host                 atan2 + first-fit clustering of deskew.cpp:158-201 (host libm, as in the reference), matrices of
//                        getRotationMatrix2D / warpAffine's inversion
//   k_warp<CH>           warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) between the two bitwise_not of rotate.cpp:61-70: 10-bit
//                        fixed-point coordinates from float64 (one rounding per operation), 5-bit bilinear weights,
//                        (sum + 2^14) >> 15.  k_rot90<CH>: the transpose/flip branches (:38-58).
// findOrientation (deskew.cpp:238, Leptonica pixOrientDetectDwa) is a no-op for the 1-channel page prl::deskew hands it
// (oracle header); prl::findOrientation of the host layer returns that 0 (prl_host.cpp), no kernel is built for it.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {
namespace {

constexpr int kNumAngle = 180;

// ---- Otsu --------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_hist(PageSet src, int width, int height, unsigned* __restrict__ hist)
{
    __shared__ unsigned h[256];
    const int page = blockIdx.z, t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const uint8_t* base = src.page(page);
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const uint8_t* row = base + (size_t)y * src.step;
        for (int x = (blockIdx.x * 256 + t) * 4; x < width; x += gridDim.x * 1024) {
            const int n = min(4, width - x);
            for (int i = 0; i < n; ++i) atomicAdd(&h[row[x + i]], 1u);
        }
    }
    __syncthreads();
    if (h[t]) atomicAdd(&hist[(size_t)page * 256 + t], h[t]);
}

// getThreshVal_Otsu_8u [upstream]: the float64 scan over the 256 bins, one thread per page.
__global__ void k_otsu(const unsigned* __restrict__ hist, int width, int height, int n_pages, int* __restrict__ thr)
{
    const int page = blockIdx.x * blockDim.x + threadIdx.x;
    if (page >= n_pages) return;
    const unsigned* h = hist + (size_t)page * 256;
    double mu = 0;
    const double scale = 1. / ((double)width * height);
    for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; ++i) {
        const double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        if (fmin(q1, q2) < (double)FLT_EPSILON || fmax(q1, q2) > 1. - (double)FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    thr[page] = (int)max_val;
}

// ---- HoughLinesP stage 1 -------------------------------------------------------------------------------------------

// one wavefront per row: mask byte = (p <= thr), row_count = number of set pixels
__global__ void __launch_bounds__(64) k_dark_mask(PageSet src, int width, int height, const int* __restrict__ thr,
                                                  uint8_t* __restrict__ mask, size_t mask_page, unsigned* __restrict__ row_count)
{
    const int page = blockIdx.y, y = blockIdx.x, lane = threadIdx.x;
    const uint8_t* row = src.page(page) + (size_t)y * src.step;
    uint8_t* m = mask + (size_t)page * mask_page + (size_t)y * width;
    const int t = thr[page];
    unsigned cnt = 0;
    for (int x = lane; x < width; x += 64) {
        const uint8_t v = row[x] <= t;
        m[x] = v;
        cnt += v;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) row_count[(size_t)page * height + y] = cnt;
}

// one workgroup per page: row_off[y] = sum of row_count[0..y), count[page] = total
__global__ void __launch_bounds__(256) k_row_offsets(int height, const unsigned* __restrict__ row_count,
                                                     unsigned* __restrict__ row_off, unsigned* __restrict__ count)
{
    __shared__ unsigned part[256];
    const int page = blockIdx.x, t = threadIdx.x;
    const unsigned* rc = row_count + (size_t)page * height;
    unsigned* ro = row_off + (size_t)page * height;
    const int per = (height + 255) / 256, y0 = t * per, y1 = min(height, y0 + per);
    unsigned s = 0;
    for (int y = y0; y < y1; ++y) s += rc[y];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned run = 0;
        for (int i = 0; i < 256; ++i) {
            const unsigned v = part[i];
            part[i] = run;
            run += v;
        }
        count[page] = run;
    }
    __syncthreads();
    unsigned run = part[t];
    for (int y = y0; y < y1; ++y) {
        ro[y] = run;
        run += rc[y];
    }
}

// one wavefront per row: the row's set pixels, in x order, to nz[nz_off[page] + row_off[y] ...]
__global__ void __launch_bounds__(64) k_collect(int width, int height, const uint8_t* __restrict__ mask, size_t mask_page,
                                                const unsigned* __restrict__ row_off, const unsigned long long* __restrict__ nz_off,
                                                unsigned* __restrict__ nz)
{
    const int page = blockIdx.y, y = blockIdx.x, lane = threadIdx.x;
    const uint8_t* m = mask + (size_t)page * mask_page + (size_t)y * width;
    unsigned* out = nz + nz_off[page] + row_off[(size_t)page * height + y];
    unsigned base = 0;
    for (int x0 = 0; x0 < width; x0 += 64) {
        const int x = x0 + lane;
        const bool set = x < width && m[x] != 0;
        const unsigned long long b = __ballot(set);
        if (set) out[base + __popcll(b & ((1ull << lane) - 1))] = (unsigned)x | ((unsigned)y << 16);
        base += __popcll(b);
    }
}

// ---- HoughLinesP stage 2 -------------------------------------------------------------------------------------------

struct PphtArgs {
    int width, height, numrho, threshold, line_length, line_gap;
    uint8_t* mask; size_t mask_page;
    unsigned* nz; const unsigned long long* nz_off; const unsigned* count;
    int* accum;                 // per page kNumAngle * numrho, zeroed
    const float* ttab;          // kNumAngle x {cos, sin}
    int* lines; const unsigned long long* lines_off; const unsigned* lines_cap; unsigned* n_lines;
    int prio;                   // raise the wavefront priority (PRL_HIP_PPHT_PRIO)
    const int* page_list;       // workgroup b takes page page_list[b], its accumulator is the b-th (null: page b)
    unsigned long long* prof;   // hooks build, PRL_HIP_PPHT_PROF=1: 16 counters per page (cycles per phase, event counts); else null
};

// Phase accounting of k_ppht_mw (hooks build only: the product kernel carries none of it).  Wavefront 0 / lane 0 adds the
// shader-clock cycles since the previous mark to the phase that just ended.
#ifdef PRL_TEST_HOOKS
#define PPHT_MARK(slot)                                                                  \
    do {                                                                                 \
        if (a.prof && wv == 0) {                                                         \
            const unsigned long long now_ = __builtin_readcyclecounter();                \
            if (lane == 0) a.prof[(size_t)page * 16 + (slot)] += now_ - prof_t;          \
            prof_t = now_;                                                               \
        }                                                                                \
    } while (0)
#define PPHT_COUNT(slot, n)                                                              \
    do {                                                                                 \
        if (a.prof && wv == 0 && lane == 0) a.prof[(size_t)page * 16 + (slot)] += (n);   \
    } while (0)
#else
#define PPHT_MARK(slot) do { } while (0)
#define PPHT_COUNT(slot, n) do { } while (0)
#endif

__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }

__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

// Steps [0, ...) of a walk x += dx, y += dy from (x0, y0): pixel of step s.
__device__ __forceinline__ void step_pixel(int xflag, unsigned x0, unsigned y0, int dx, int dy, unsigned s, int* j1, int* i1)
{
    const int x = (int)(x0 + s * (unsigned)dx), y = (int)(y0 + s * (unsigned)dy);  // wraps like the reference's repeated adds
    if (xflag) { *j1 = x; *i1 = y >> 16; }
    else { *j1 = x >> 16; *i1 = y; }
}

// max over the 64 lanes of a signed value (DPP row shifts + row broadcasts; the result is uniform)
__device__ __forceinline__ int wave_max_i32(int x)
{
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x112, 0xf, 0xf, false));  // row_shr:2
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x114, 0xf, 0xf, false));  // row_shr:4
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x118, 0xf, 0xf, false));  // row_shr:8
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(x, 63);
}

#ifndef PRL_PPHT_BLK
#define PRL_PPHT_BLK 32
#endif
// Points per block of k_ppht.  32: 96 returned counts in registers (136 VGPRs); 64 (234 VGPRs) measured 3 % slower on 256+
// pages - the batch keeps the memory system busy either way - and 24 / 16 the same as 32 (profiles/r02/chain_overlap.txt).
constexpr int kBlk = PRL_PPHT_BLK;

// A vote.  A lane is the only one that ever touches its cells, but a narrower scope buys nothing: wavefront, workgroup and agent
// scope compile to the same `global_atomic_add ... sc0`; where it executes is decided by the memory type (hipMalloc: behind the
// L2s, profiles/r02/pmc_ppht.txt).
__device__ __forceinline__ int vote(int* cell, int d) { return __hip_atomic_fetch_add(cell, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Cells hold count + 2^31 (the accumulator is filled with 0x80000000): a vote's returned value gives the count by flipping the top
// bit, and the un-votes of a good line can be taken from TWO adjacent cells with one 64-bit subtraction - the low half never
// borrows from the high one, because a biased cell is never smaller than what is subtracted from it.
constexpr unsigned kAccBias = 0x80000000u;
static_assert(kNumAngle % 2 == 0, "a page's accumulator must be a whole number of 8-byte cell pairs (Unvotes)");
__device__ __forceinline__ int count_of(int stored) { return (int)((unsigned)stored ^ kAccBias); }

// The un-votes of one lane's angle along a line walk.  For a fixed angle the cell index r = round(x cos + y sin) is monotone
// along the walk and moves by at most sqrt(2) per step, so consecutive erased pixels hit the same cell or its neighbour: the
// decrements are collected per aligned PAIR of cells and leave as one global_atomic_sub_x2 when the walk moves on - about a
// third of the atomics of one decrement per pixel and angle (the transform is bound by the rate of memory-side atomics:
// 2.1e10 per second for the whole chip), with identical cell values at every later vote.
struct Unvotes {
    int pair = -1;
    unsigned lo = 0, hi = 0;
    __device__ __forceinline__ void flush(unsigned long long* acc64)
    {
        if (pair >= 0)
            (void)__hip_atomic_fetch_sub(acc64 + pair, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        pair = -1;
        lo = hi = 0;
    }
    __device__ __forceinline__ void add(unsigned long long* acc64, int idx)   // idx: cell index from the page's accumulator base
    {
        const int p = idx >> 1;
        if (p != pair) {
            flush(acc64);
            pair = p;
        }
        if (idx & 1) ++hi;
        else ++lo;
    }
};

// HoughLinesP stage 2, one wavefront per page, BLOCKS OF kBlk POINTS (round 2, second version).  The first version walked the
// points one by one: list fetch -> mask byte -> 180 voting atomics -> decision, three dependent memory round trips per
// point (1.9 us).  Two facts allow overlapping them without changing a single result:
//  * the visiting order does not depend on the data (cv::RNG draws idx_t = next() % (N - t) and the list swap
//    nz[idx_t] = nz[N - t - 1] happens whatever the point does), so kBlk steps of it are taken at once: lane L fetches the list
//    entries of step t0 + L, and the swaps of the earlier steps of the same block are applied to its values in registers
//    (a kBlk-step loop of readlane / compare / select), duplicates resolved so that memory ends in the sequential state;
//  * a vote only matters when its cell reaches the threshold, which a fraction of a percent of the points do: the votes of
//    all points of the block are ISSUED back to back (atomics with return into 3 kBlk registers; a lane owns its angles' rows, so
//    successive points hitting one cell arrive in order) and then RETIRED in order; the first point that triggers a line has
//    the votes of the younger points taken back, its line is walked exactly as before, and the rest of the block starts
//    over (masks re-read: the walk may have erased some of them).
// Up to 63 vector memory instructions are in flight per wavefront (the vmcnt counter), i.e. about 21 points.
__global__ void __launch_bounds__(64) k_ppht(PphtArgs a)
{
    // One latency-bound wavefront per page that issues little: in the chain it shares its SIMD with the NL-means wavefronts of
    // the previous pass (glue.hip), which saturate the vector ALU - let the arbiter pick this one first.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const int page = a.page_list ? a.page_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    const int W = a.width, H = a.height, numrho = a.numrho;
    volatile uint8_t* mask = a.mask + (size_t)page * a.mask_page;
    volatile unsigned* nz = a.nz + a.nz_off[page];
    int* accum = a.accum + (size_t)blockIdx.x * kNumAngle * numrho;
    int* lines = a.lines + a.lines_off[page] * 4;
    const unsigned cap = a.lines_cap[page];
    float tc[3], ts[3];
    int* arow[3];
    int row0[3];   // index of cell r = 0 of the lane's three angle rows
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = min(lane + 64 * q, kNumAngle - 1);
        tc[q] = a.ttab[2 * n];
        ts[q] = a.ttab[2 * n + 1];
        arow[q] = accum + (size_t)n * numrho + (numrho - 1) / 2;
        row0[q] = n * numrho + (numrho - 1) / 2;
    }
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(accum);
    const bool has2 = lane + 128 < kNumAngle;
    unsigned long long rng = ~0ull;
    unsigned n_lines = 0;
    const unsigned N = a.count[page];

    // One detected line: walk both ways from (j, i) along angle max_n, clear the mask, take votes back (as the first version).
    auto process_line = [&](int j, int i, int max_n) {
        const float fa = -a.ttab[2 * max_n + 1], fb = a.ttab[2 * max_n];
        unsigned x0 = (unsigned)j, y0 = (unsigned)i;
        int dx0, dy0, xflag;
        if (fabs((double)fa) > fabs((double)fb)) {
            xflag = 1;
            dx0 = fa > 0 ? 1 : -1;
            dy0 = __double2int_rn((double)(fb * 65536.f) / fabs((double)fa));
            y0 = (y0 << 16) + (1u << 15);
        } else {
            xflag = 0;
            dy0 = fb > 0 ? 1 : -1;
            dx0 = __double2int_rn((double)(fa * 65536.f) / fabs((double)fb));
            x0 = (x0 << 16) + (1u << 15);
        }
        // first walk, both directions: the step of the last set pixel before the border or a gap above line_gap
        unsigned end_step[2];
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            unsigned endk = 0;
            int gap = 0;  // consecutive cleared pixels so far
            bool stop = false;
            for (unsigned base = 0; !stop; base += 64) {
                int j1, i1;
                step_pixel(xflag, x0, y0, dx, dy, base + lane, &j1, &i1);
                const bool inb = j1 >= 0 && j1 < W && i1 >= 0 && i1 < H;
                const bool set = inb && mask[(size_t)i1 * W + j1] != 0;
                const unsigned long long oob = __ballot(!inb);
                unsigned long long nzb = __ballot(set);
                const int limit = oob ? __ffsll((long long)oob) - 1 : 64;  // steps [0, limit) are inside the image
                int prev = -1 - gap;                                        // position of the last set pixel (virtual)
                while (nzb) {
                    const int q = __ffsll((long long)nzb) - 1;
                    nzb &= nzb - 1;
                    if (q >= limit || q - prev - 1 > a.line_gap) { stop = true; break; }
                    endk = base + (unsigned)q;
                    prev = q;
                }
                if (!stop) {
                    if (limit - 1 - prev > a.line_gap || limit < 64) stop = true;
                    else gap = 63 - prev;
                }
            }
            end_step[k] = endk;
        }
        int ex[2], ey[2];
        step_pixel(xflag, x0, y0, dx0, dy0, end_step[0], &ex[0], &ey[0]);
        step_pixel(xflag, x0, y0, -dx0, -dy0, end_step[1], &ex[1], &ey[1]);
        const bool good_line = abs(ex[1] - ex[0]) >= a.line_length || abs(ey[1] - ey[0]) >= a.line_length;
        // second walk: clear the set pixels up to the line ends; a good line takes their votes back
        Unvotes uv[3];
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (unsigned base = (unsigned)k; base <= end_step[k]; base += 64) {  // (step 0 was cleared by k = 0)
                const unsigned st = base + lane;
                int j1, i1;
                step_pixel(xflag, x0, y0, dx, dy, st, &j1, &i1);
                bool set = false;
                if (st <= end_step[k]) {
                    set = mask[(size_t)i1 * W + j1] != 0;
                    if (set) mask[(size_t)i1 * W + j1] = 0;
                }
                unsigned long long nzb = __ballot(set);
                if (good_line) {
                    while (nzb) {
                        const int q = __ffsll((long long)nzb) - 1;
                        nzb &= nzb - 1;
                        int jq, iq;
                        step_pixel(xflag, x0, y0, dx, dy, base + (unsigned)q, &jq, &iq);
#pragma unroll
                        for (int qq = 0; qq < 3; ++qq)
                            if (qq < 2 || has2) uv[qq].add(acc64, row0[qq] + cv_round_f((float)jq * tc[qq] + (float)iq * ts[qq]));
                    }
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) uv[qq].flush(acc64);   // before any later vote reads these cells
        if (good_line) {
            if (lane == 0 && n_lines < cap) {
                lines[4 * n_lines] = ex[0];
                lines[4 * n_lines + 1] = ey[0];
                lines[4 * n_lines + 2] = ex[1];
                lines[4 * n_lines + 3] = ey[1];
            }
            ++n_lines;
        }
    };

    for (unsigned t0 = 0; t0 < N; t0 += kBlk) {
        const unsigned nb = min((unsigned)kBlk, N - t0), c0 = N - t0;
        // the next nb outputs of cv::RNG, one per lane
        unsigned rv = 0;
        for (unsigned L = 0; L < nb; ++L) {
            rng = (unsigned long long)(unsigned)rng * 4164903690ull + (rng >> 32);
            if ((unsigned)lane == L) rv = (unsigned)rng;
        }
        const bool act = (unsigned)lane < nb;
        const unsigned cnt_l = act ? c0 - (unsigned)lane : 1u;  // list length at this lane's step
        const unsigned idx = rv % cnt_l, lastpos = cnt_l - 1u;
        unsigned rp = 0, rl = 0;
        if (act) {
            rp = nz[idx];
            rl = nz[lastpos];
        }
        // swaps of the earlier steps of this block, applied in registers; `skip`: steps whose slot a later step writes again
        unsigned long long skip = 0;
        for (unsigned st = 0; st + 1 < nb; ++st) {
            const unsigned is = (unsigned)__builtin_amdgcn_readlane((int)idx, (int)st);
            const unsigned ls = (unsigned)__builtin_amdgcn_readlane((int)rl, (int)st);
            const bool later = act && (unsigned)lane > st;
            const bool c1 = later && idx == is, c2 = later && lastpos == is;
            if (c1) rp = ls;
            if (c2) rl = ls;
            if (__ballot(c1)) skip |= 1ull << st;
        }
        if (act && !((skip >> lane) & 1ull)) nz[idx] = rl;   // memory ends in the sequential state
        const unsigned pt = rp;                               // this lane's point: x | y << 16

        unsigned start = 0;
        while (start < nb) {
            // mask bytes of the points not yet retired (a line walked meanwhile may have erased some)
            unsigned m = 0;
            if (act && (unsigned)lane >= start) m = mask[(size_t)(pt >> 16) * W + (pt & 0xffffu)];
            const unsigned long long vm = __ballot(m != 0);
            if (!vm) break;
            // issue: every point's votes, back to back; v[p][q] = the count before this vote
            int v[kBlk][3];
#pragma unroll
            for (int p = 0; p < kBlk; ++p) {
                if ((vm >> p) & 1ull) {
                    const unsigned q = (unsigned)__builtin_amdgcn_readlane((int)pt, p);
                    const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq)
                        if (qq < 2 || has2) v[p][qq] = vote(arow[qq] + cv_round_f(fj * tc[qq] + fi * ts[qq]), 1);
                }
            }
            // retire in order: key = count * 256 + (255 - angle), signed (counts go negative: a good line takes back the votes
            // of points that have not voted yet, as in the reference): the largest count, the first angle among equals
            int trig = -1, trig_n = 0;
#pragma unroll
            for (int p = 0; p < kBlk; ++p) {
                if (((vm >> p) & 1ull) && trig < 0) {
                    int key = INT_MIN;
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq)
                        if (qq < 2 || has2) key = max(key, (count_of(v[p][qq]) + 1) * 256 + (255 - (lane + 64 * qq)));
                    key = wave_max_i32(key);
                    if ((key >> 8) >= a.threshold) {
                        trig = p;
                        trig_n = 255 - (key & 255);
                    }
                }
            }
            if (trig < 0) break;  // every vote of the block stands
            // the younger points voted on speculation: take their votes back, walk the line, start over behind it
            for (unsigned p = (unsigned)trig + 1; p < nb; ++p) {
                if ((vm >> p) & 1ull) {
                    const unsigned q = (unsigned)__builtin_amdgcn_readlane((int)pt, (int)p);
                    const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq)
                        if (qq < 2 || has2) vote(arow[qq] + cv_round_f(fj * tc[qq] + fi * ts[qq]), -1);
                }
            }
            const unsigned tq = (unsigned)__builtin_amdgcn_readlane((int)pt, trig);
            process_line((int)(tq & 0xffffu), (int)(tq >> 16), trig_n);
            start = (unsigned)trig + 1;
        }
    }
    if (lane == 0) a.n_lines[page] = n_lines;
}

// The same transform with THREE wavefronts per page (small batches: a single wavefront per page is bound by its own memory
// round trips - 0.7 s for an A4 page whatever the batch size up to ~64 pages - while the chip idles).  Wavefront v, lane l
// owns angle 64 v + l: a point's 180 votes are one instruction per wavefront, a block's votes are all in flight at once, and
// un-votes and roll-backs split three ways.  Everything that decides the visiting order (cv::RNG, the list swaps, the masks)
// is computed by all three wavefronts identically; what WRITES shared state is done by wavefront 0 alone (list stores, mask
// clearing, the segment list) and handed on through LDS / a barrier: the per-point maxima (keys, double-buffered), the set
// pixels of the clearing walk (one ballot per 64 steps).  Results are identical to k_ppht's by construction and by
// tests/test_deskew_gpu.py, which runs its cases through both kernels (PRL_HIP_PPHT_MW=0 in a child process).
constexpr int kMwWaves = 3;
#ifndef PRL_PPHT_BLK_MW
#define PRL_PPHT_BLK_MW 32
#endif
constexpr int kBlkMw = PRL_PPHT_BLK_MW;
constexpr int kPphtMwMaxPages = 1 << 30;   // never slower than one wavefront per page (1 .. 512 pages measured: DESIGN.md 4.10)
constexpr int kMwMaxGroups = 2 * (32768 / 64 + 2);
__global__ void __launch_bounds__(64 * kMwWaves) k_ppht_mw(PphtArgs a)
{
    __shared__ int s_keys[2][kMwWaves][kBlkMw];
    __shared__ unsigned long long s_ballot[kMwMaxGroups];
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const int page = a.page_list ? a.page_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W = a.width, H = a.height, numrho = a.numrho;
    volatile uint8_t* mask = a.mask + (size_t)page * a.mask_page;
    volatile unsigned* nz = a.nz + a.nz_off[page];
    int* accum = a.accum + (size_t)blockIdx.x * kNumAngle * numrho;
    int* lines = a.lines + a.lines_off[page] * 4;
    const unsigned cap = a.lines_cap[page];
    const int ang = wv * 64 + lane;
    const bool has = ang < kNumAngle;
    const int angc = min(ang, kNumAngle - 1);
    const float tc = a.ttab[2 * angc], ts = a.ttab[2 * angc + 1];
    int* arow = accum + (size_t)angc * numrho + (numrho - 1) / 2;
    const int row0 = angc * numrho + (numrho - 1) / 2;   // index of this lane's cell r = 0
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(accum);
    unsigned long long rng = ~0ull;
    unsigned n_lines = 0;
    const unsigned N = a.count[page];
    int kbuf = 0;
#ifdef PRL_TEST_HOOKS
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif

    auto process_line = [&](int j, int i, int max_n) {
        const float fa = -a.ttab[2 * max_n + 1], fb = a.ttab[2 * max_n];
        unsigned x0 = (unsigned)j, y0 = (unsigned)i;
        int dx0, dy0, xflag;
        if (fabs((double)fa) > fabs((double)fb)) {
            xflag = 1;
            dx0 = fa > 0 ? 1 : -1;
            dy0 = __double2int_rn((double)(fb * 65536.f) / fabs((double)fa));
            y0 = (y0 << 16) + (1u << 15);
        } else {
            xflag = 0;
            dy0 = fb > 0 ? 1 : -1;
            dx0 = __double2int_rn((double)(fa * 65536.f) / fabs((double)fb));
            x0 = (x0 << 16) + (1u << 15);
        }
        // first walk (read only): every wavefront for itself, the results are identical
        unsigned end_step[2];
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            unsigned endk = 0;
            int gap = 0;
            bool stop = false;
            for (unsigned base = 0; !stop; base += 64) {
                int j1, i1;
                step_pixel(xflag, x0, y0, dx, dy, base + lane, &j1, &i1);
                const bool inb = j1 >= 0 && j1 < W && i1 >= 0 && i1 < H;
                const bool set = inb && mask[(size_t)i1 * W + j1] != 0;
                const unsigned long long oob = __ballot(!inb);
                unsigned long long nzb = __ballot(set);
                const int limit = oob ? __ffsll((long long)oob) - 1 : 64;
                int prev = -1 - gap;
                while (nzb) {
                    const int q = __ffsll((long long)nzb) - 1;
                    nzb &= nzb - 1;
                    if (q >= limit || q - prev - 1 > a.line_gap) { stop = true; break; }
                    endk = base + (unsigned)q;
                    prev = q;
                }
                if (!stop) {
                    if (limit - 1 - prev > a.line_gap || limit < 64) stop = true;
                    else gap = 63 - prev;
                }
            }
            end_step[k] = endk;
        }
        int ex[2], ey[2];
        step_pixel(xflag, x0, y0, dx0, dy0, end_step[0], &ex[0], &ey[0]);
        step_pixel(xflag, x0, y0, -dx0, -dy0, end_step[1], &ex[1], &ey[1]);
        const bool good_line = abs(ex[1] - ex[0]) >= a.line_length || abs(ey[1] - ey[0]) >= a.line_length;
        PPHT_MARK(3);   // read-only walk
        PPHT_COUNT(9, (end_step[0] + end_step[1]) / 64 + 2);
        __syncthreads();  // nobody clears a pixel before everybody has finished the read-only walk
        // second walk: wavefront 0 clears the set pixels and publishes which ones they were
        if (wv == 0) {
            int gidx = 0;
            for (int k = 0; k < 2; ++k) {
                const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
                for (unsigned base = (unsigned)k; base <= end_step[k]; base += 64, ++gidx) {  // (step 0 was cleared by k = 0)
                    const unsigned st = base + lane;
                    int j1, i1;
                    step_pixel(xflag, x0, y0, dx, dy, st, &j1, &i1);
                    bool set = false;
                    if (st <= end_step[k]) {
                        set = mask[(size_t)i1 * W + j1] != 0;
                        if (set) mask[(size_t)i1 * W + j1] = 0;
                    }
                    const unsigned long long nzb = __ballot(set);
                    if (lane == 0) s_ballot[gidx] = nzb;
                }
            }
        }
        __syncthreads();
        PPHT_MARK(4);   // clearing walk
        if (good_line) {  // a good line takes the votes of its pixels back: every wavefront its own angles
            Unvotes uv;
            int gidx = 0;
            for (int k = 0; k < 2; ++k) {
                const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
                for (unsigned base = (unsigned)k; base <= end_step[k]; base += 64, ++gidx) {
                    unsigned long long nzb = s_ballot[gidx];
                    PPHT_COUNT(10, __popcll(nzb));
                    while (nzb) {
                        const int q = __ffsll((long long)nzb) - 1;
                        nzb &= nzb - 1;
                        int jq, iq;
                        step_pixel(xflag, x0, y0, dx, dy, base + (unsigned)q, &jq, &iq);
                        if (has) uv.add(acc64, row0 + cv_round_f((float)jq * tc + (float)iq * ts));
                    }
                }
            }
            uv.flush(acc64);   // before any later vote reads these cells
            if (wv == 0 && lane == 0 && n_lines < cap) {
                lines[4 * n_lines] = ex[0];
                lines[4 * n_lines + 1] = ey[0];
                lines[4 * n_lines + 2] = ex[1];
                lines[4 * n_lines + 3] = ey[1];
            }
            ++n_lines;
        }
        __syncthreads();  // s_ballot is free again; the cleared pixels are visible to everybody's next mask reads
        PPHT_MARK(5);   // un-votes (+ the segment store)
        PPHT_COUNT(8, good_line ? 1 : 0);
    };

    for (unsigned t0 = 0; t0 < N; t0 += kBlkMw) {
        const unsigned nb = min((unsigned)kBlkMw, N - t0), c0 = N - t0;
        unsigned rv = 0;
        for (unsigned L = 0; L < nb; ++L) {
            rng = (unsigned long long)(unsigned)rng * 4164903690ull + (rng >> 32);
            if ((unsigned)lane == L) rv = (unsigned)rng;
        }
        const bool act = (unsigned)lane < nb;
        const unsigned cnt_l = act ? c0 - (unsigned)lane : 1u;
        const unsigned idx = rv % cnt_l, lastpos = cnt_l - 1u;
        unsigned rp = 0, rl = 0;
        if (act) {
            rp = nz[idx];
            rl = nz[lastpos];
        }
        unsigned long long skip = 0;
        for (unsigned st = 0; st + 1 < nb; ++st) {
            const unsigned is = (unsigned)__builtin_amdgcn_readlane((int)idx, (int)st);
            const unsigned ls = (unsigned)__builtin_amdgcn_readlane((int)rl, (int)st);
            const bool later = act && (unsigned)lane > st;
            const bool c1 = later && idx == is, c2 = later && lastpos == is;
            if (c1) rp = ls;
            if (c2) rl = ls;
            if (__ballot(c1)) skip |= 1ull << st;
        }
        __syncthreads();  // every wavefront has read this block's list entries before the list moves on.  All three store the
        // same values: a wavefront's own store then orders its own reads of the next block (a block without a live point has
        // no further barrier)
        if (act && !((skip >> lane) & 1ull)) nz[idx] = rl;
        const unsigned pt = rp;
        PPHT_MARK(0);   // block prologue: RNG, list fetch, swaps
        PPHT_COUNT(6, 1);

        unsigned start = 0;
        while (start < nb) {
            unsigned m = 0;
            if (act && (unsigned)lane >= start) m = mask[(size_t)(pt >> 16) * W + (pt & 0xffffu)];
            const unsigned long long vm = __ballot(m != 0);
            if (!vm) { PPHT_MARK(1); break; }
            PPHT_COUNT(11, __popcll(vm));
            int v[kBlkMw];
#pragma unroll
            for (int p = 0; p < kBlkMw; ++p) {
                if ((vm >> p) & 1ull) {
                    const unsigned q = (unsigned)__builtin_amdgcn_readlane((int)pt, p);
                    const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
                    if (has) v[p] = vote(arow + cv_round_f(fj * tc + fi * ts), 1);
                }
            }
            // this wavefront's maximum per point into lane p, then the three of them through LDS
            int mine = INT_MIN;
#pragma unroll
            for (int p = 0; p < kBlkMw; ++p) {
                if ((vm >> p) & 1ull) {
                    const int key = wave_max_i32(has ? (count_of(v[p]) + 1) * 256 + (255 - ang) : INT_MIN);
                    if (lane == p) mine = key;
                }
            }
            if (lane < kBlkMw) s_keys[kbuf][wv][lane] = mine;
            __syncthreads();
            int kk = INT_MIN;
            if (lane < kBlkMw) {
#pragma unroll
                for (int u = 0; u < kMwWaves; ++u) kk = max(kk, s_keys[kbuf][u][lane]);
            }
            kbuf ^= 1;
            const unsigned long long tb = __ballot(lane < kBlkMw && ((vm >> lane) & 1ull) && (kk >> 8) >= a.threshold);
            PPHT_MARK(1);   // mask bytes, votes issued and retired
            if (!tb) break;  // every vote of the block stands
            PPHT_COUNT(7, 1);
            const int trig = __ffsll((long long)tb) - 1;
            const int trig_n = 255 - (__builtin_amdgcn_readlane(kk, trig) & 255);
            for (unsigned p = (unsigned)trig + 1; p < nb; ++p) {
                if ((vm >> p) & 1ull) {
                    const unsigned q = (unsigned)__builtin_amdgcn_readlane((int)pt, (int)p);
                    const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
                    if (has) vote(arow + cv_round_f(fj * tc + fi * ts), -1);
                }
            }
            const unsigned tq = (unsigned)__builtin_amdgcn_readlane((int)pt, trig);
            PPHT_MARK(2);   // roll-back of the younger points' votes
            process_line((int)(tq & 0xffffu), (int)(tq >> 16), trig_n);
            start = (unsigned)trig + 1;
        }
    }
    if (wv == 0 && lane == 0) a.n_lines[page] = n_lines;
}

// ---- rotate --------------------------------------------------------------------------------------------------------

struct WarpPage {
    double M[6];   // the inverted matrix warpAffine works with
    int kind;      // 0 warp, 1/2/3 = 90/180/270, 4 = copy (deskew without an angle)
    int ow, oh;
};

// 8 bytes at any alignment (global memory takes unaligned dword accesses on gfx9+)
__device__ __forceinline__ uint2 load8u(const uint8_t* p)
{
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ unsigned byte_at(uint2 v, int i) { return ((i < 4 ? v.x : v.y) >> (8 * (i & 3))) & 0xffu; }

// A workgroup produces 256 consecutive pixels of kWarpRows output rows; the result bytes go through LDS so that a row segment is
// stored as dwords (byte stores of CH-byte pixels cost three to four instructions per pixel).  The column terms of the source
// coordinates (warpAffine's adelta / bdelta tables, float64) are formed once per thread and reused for its rows, the row terms
// once per row by one lane.  Pages of kind 1 / 3 (quarter turns) are left to k_rot90.
constexpr int kWarpRows = 4;
template <int CH>
__global__ void __launch_bounds__(256) k_warp(PageSet src, PageSetOut dst, int width, int height, const WarpPage* __restrict__ wp)
{
    __shared__ __attribute__((aligned(16))) uint8_t seg[256 * CH + 16];
    __shared__ int rowXY[kWarpRows][2];
    const int page = blockIdx.z, y0 = blockIdx.y * kWarpRows, x0 = blockIdx.x * 256, x = x0 + (int)threadIdx.x;
    const WarpPage& p = wp[page];
    if (p.kind == 1 || p.kind == 3 || x0 >= p.ow || y0 >= p.oh) return;
    const uint8_t* s = src.page(page);
    int adelta = 0, bdelta = 0;
    if (p.kind == 0) {
        if (threadIdx.x < kWarpRows) {
            const int y = y0 + (int)threadIdx.x;
            rowXY[threadIdx.x][0] = __double2int_rn((p.M[1] * y + p.M[2]) * 1024) + 16;
            rowXY[threadIdx.x][1] = __double2int_rn((p.M[4] * y + p.M[5]) * 1024) + 16;
        }
        adelta = __double2int_rn(p.M[0] * x * 1024);
        bdelta = __double2int_rn(p.M[3] * x * 1024);
        __syncthreads();
    }
    const int nrows = min(kWarpRows, p.oh - y0);
    for (int ry = 0; ry < nrows; ++ry) {
        const int y = y0 + ry;
        unsigned res[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) res[c] = 0;
        if (x < p.ow) {
            if (p.kind != 0) {  // 2: half turn; 4: copy
                const int sx = p.kind == 2 ? width - 1 - x : x, sy = p.kind == 2 ? height - 1 - y : y;
                const uint8_t* q = s + (size_t)sy * src.step + (size_t)sx * CH;
#pragma unroll
                for (int c = 0; c < CH; ++c) res[c] = q[c];
            } else {
                const int X = (rowXY[ry][0] + adelta) >> 5, Y = (rowXY[ry][1] + bdelta) >> 5;
                const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
                const int fx = X & 31, fy = Y & 31;
                const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
                const bool x0in = sx >= 0 && sx < width, x1in = sx + 1 >= 0 && sx + 1 < width;
                const bool y0in = sy >= 0 && sy < height, y1in = sy + 1 >= 0 && sy + 1 < height;
                const uint8_t* r0 = s + (size_t)sy * src.step + (size_t)sx * CH;  // only dereferenced where the flags allow
                const uint8_t* r1 = r0 + src.step;
                // both taps of a row are 2 * CH <= 8 consecutive bytes: one 8-byte fetch per row when all four taps are inside
                // and the fetch cannot run past the page's last row
                const bool wide = x0in && x1in && y0in && y1in && (sy + 2 < height || sx * CH + 8 <= (int)src.step);
                if (wide) {
                    const uint2 a = load8u(r0), b = load8u(r1);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        // the source is cv::bitwise_not(input) and the weights add up to 2^15:
                        // sum w (255 - v) = 255 * 2^15 - sum w v
                        const int wv = (int)byte_at(a, c) * w00 + (int)byte_at(a, CH + c) * w01 + (int)byte_at(b, c) * w10 +
                                       (int)byte_at(b, CH + c) * w11;
                        res[c] = (unsigned)(255 - ((255 * 32768 - wv + (1 << 14)) >> 15));
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        // outside the (inverted) source the border value 0
                        const int v00 = (x0in && y0in) ? 255 - r0[c] : 0, v01 = (x1in && y0in) ? 255 - r0[CH + c] : 0;
                        const int v10 = (x0in && y1in) ? 255 - r1[c] : 0, v11 = (x1in && y1in) ? 255 - r1[CH + c] : 0;
                        res[c] = (unsigned)(255 - ((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15));
                    }
                }
            }
        }
        // the segment's bytes: [head up to the first 4-byte boundary of the destination][dwords][tail]
        uint8_t* d0 = dst.page(page) + (size_t)y * dst.step + (size_t)x0 * CH;
        const int nbytes = min(256, p.ow - x0) * CH;
        const int head = min(nbytes, (int)((4 - ((size_t)d0 & 3)) & 3));
        // the LDS image is shifted so that destination-aligned dwords are LDS-aligned dwords
        const int shift = (4 - head) & 3;
        if (ry) __syncthreads();  // the previous row's segment has been stored
#pragma unroll
        for (int c = 0; c < CH; ++c) seg[shift + (int)threadIdx.x * CH + c] = (uint8_t)res[c];
        __syncthreads();
        if ((int)threadIdx.x < head) d0[threadIdx.x] = seg[shift + threadIdx.x];
        const int ndw = (nbytes - head) / 4;
        const unsigned* sw = reinterpret_cast<const unsigned*>(seg + shift + head);
        unsigned* dw = reinterpret_cast<unsigned*>(d0 + head);
        for (int i = threadIdx.x; i < ndw; i += 256) dw[i] = sw[i];
        const int tail0 = head + ndw * 4;
        if ((int)threadIdx.x < nbytes - tail0) d0[tail0 + threadIdx.x] = seg[shift + tail0 + threadIdx.x];
    }
}

// Quarter turns (rotate.cpp:38-58): out(y, x) = in(height - 1 - x, y) for 90, in(x, width - 1 - y) for 270.  A 32 x 32 tile
// is read along source rows, turned in LDS and written along destination rows (the straightforward one-thread-per-pixel
// form reads a different source row in every lane: 7.0 ms for 64 A4 colour pages, 6 % of the roofline).
template <int CH>
__global__ void __launch_bounds__(256) k_rot90(PageSet src, PageSetOut dst, int width, int height, const WarpPage* __restrict__ wp)
{
    __shared__ uint8_t tile[32][32 * CH + 4];
    const int page = blockIdx.z;
    const WarpPage& p = wp[page];
    if (p.kind != 1 && p.kind != 3) return;
    const int ox0 = blockIdx.x * 32, oy0 = blockIdx.y * 32;  // output tile; output is height (cols) x width (rows)
    if (ox0 >= p.ow || oy0 >= p.oh) return;
    const uint8_t* s = src.page(page);
    // source tile: rows sy0 .. sy0+31, cols sx0 .. sx0+31 (clipped)
    const int sx0 = p.kind == 1 ? oy0 : width - 32 - oy0, sy0 = p.kind == 1 ? height - 32 - ox0 : ox0;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        const int sy = sy0 + r, sx = sx0 + c;
        if (sy >= 0 && sy < height && sx >= 0 && sx < width) {
            const uint8_t* q = s + (size_t)sy * src.step + (size_t)sx * CH;
#pragma unroll
            for (int k = 0; k < CH; ++k) tile[r][c * CH + k] = q[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int oy = i >> 5, ox = i & 31;
        const int y = oy0 + oy, x = ox0 + ox;
        if (y >= p.oh || x >= p.ow) continue;
        // kind 1: in(height-1-x, y): row index height-1-x - sy0 = 31 - ox, col y - sx0 = oy
        // kind 3: in(x, width-1-y): row x - sy0 = ox, col width-1-y - sx0 = 31 - oy
        const int r = p.kind == 1 ? 31 - ox : ox, c = p.kind == 1 ? oy : 31 - oy;
        uint8_t* d = dst.page(page) + (size_t)y * dst.step + (size_t)x * CH;
#pragma unroll
        for (int k = 0; k < CH; ++k) d[k] = tile[r][c * CH + k];
    }
}

bool eq_d(double a, double b, double delta) { return std::fabs(a - b) <= delta; }

// rotate.cpp:37-58
int rotate_kind(double angle)
{
    angle = std::fmod(angle, 360.0);
    if (eq_d(angle, 90.0, 1e-7)) return 1;
    if (eq_d(angle, 180.0, 1e-7)) return 2;
    if (eq_d(angle, 270.0, 1e-7)) return 3;
    return 0;
}

// getRotationMatrix2D((len/2, len/2), angle, 1.0) then warpAffine's inversion [upstream, see the oracle]
void rotate_matrix(int width, int height, double angle, double M[6])
{
    const int len = std::max(width, height);
    const float cx = static_cast<float>(len / 2.0), cy = static_cast<float>(len / 2.0);
    angle = std::fmod(angle, 360.0);
    angle *= 3.1415926535897932384626433832795 / 180;
    const double alpha = std::cos(angle) * 1.0, beta = std::sin(angle) * 1.0;
    M[0] = alpha; M[1] = beta; M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta; M[4] = alpha; M[5] = beta * cx + (1 - alpha) * cy;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

void fill_warp_page(int width, int height, double angle, bool copy, WarpPage* wp)
{
    wp->kind = copy ? 4 : rotate_kind(angle);
    if (wp->kind == 1 || wp->kind == 3) { wp->ow = height; wp->oh = width; }
    else if (wp->kind == 2 || wp->kind == 4) { wp->ow = width; wp->oh = height; }
    else { wp->ow = wp->oh = std::max(width, height); rotate_matrix(width, height, angle, wp->M); }
}

// deskew.cpp:158-201 over the segments of one page
// The angle vote of findAngle (deskew.cpp:158-201): every segment's atan2 joins the FIRST cluster whose representative (the angle
// that opened it) lies within 0.01 rad, else opens a new one; a cluster's tally counts the joiners only (it starts at 0, as the
// reference's does), the first cluster with the largest tally wins and its representative is returned in degrees.
double vote_angle(const int* segments, int n_segments)
{
    if (n_segments <= 0) return 0.0;
    constexpr double kClusterWidth = 0.01;
    std::vector<double> representative;   // cluster -> the angle that opened it
    std::vector<int> joiners;             // cluster -> segments that joined it afterwards
    for (int i = 0; i < n_segments; ++i) {
        const int* seg = segments + 4 * i;   // x0, y0, x1, y1
        const double theta = std::atan2((double)seg[3] - seg[1], (double)seg[2] - seg[0]);
        size_t c = 0;
        while (c < representative.size() && !eq_d(theta, representative[c], kClusterWidth)) ++c;
        if (c == representative.size()) {
            representative.push_back(theta);
            joiners.push_back(0);
        } else {
            ++joiners[c];
        }
    }
    size_t winner = 0;   // (std::max_element keeps the first of equal maxima)
    for (size_t c = 1; c < joiners.size(); ++c)
        if (joiners[c] > joiners[winner]) winner = c;
    return representative[winner] * 180 / 3.14159265358979323846;
}

int launch_warp(int channels, const PageSet& s, const PageSetOut& d, int width, int height, int n_pages, int max_ow, int max_oh,
                const WarpPage* d_wp, hipStream_t stream, bool any_quarter = true)
{
    const dim3 grid((unsigned)((max_ow + 255) / 256), (unsigned)((max_oh + kWarpRows - 1) / kWarpRows), (unsigned)n_pages);
    switch (channels) {
    case 1: hipLaunchKernelGGL(k_warp<1>, grid, dim3(256), 0, stream, s, d, width, height, d_wp); break;
    case 2: hipLaunchKernelGGL(k_warp<2>, grid, dim3(256), 0, stream, s, d, width, height, d_wp); break;
    case 3: hipLaunchKernelGGL(k_warp<3>, grid, dim3(256), 0, stream, s, d, width, height, d_wp); break;
    default: hipLaunchKernelGGL(k_warp<4>, grid, dim3(256), 0, stream, s, d, width, height, d_wp); break;
    }
    PRL_HIP_CHECK(hipGetLastError());
    if (any_quarter) {
        const dim3 g2((unsigned)((max_ow + 31) / 32), (unsigned)((max_oh + 31) / 32), (unsigned)n_pages);
        switch (channels) {
        case 1: hipLaunchKernelGGL(k_rot90<1>, g2, dim3(256), 0, stream, s, d, width, height, d_wp); break;
        case 2: hipLaunchKernelGGL(k_rot90<2>, g2, dim3(256), 0, stream, s, d, width, height, d_wp); break;
        case 3: hipLaunchKernelGGL(k_rot90<3>, g2, dim3(256), 0, stream, s, d, width, height, d_wp); break;
        default: hipLaunchKernelGGL(k_rot90<4>, g2, dim3(256), 0, stream, s, d, width, height, d_wp); break;
        }
        PRL_HIP_CHECK(hipGetLastError());
    }
    return PRL_OK;
}

size_t r256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// Totals of the Hough searches since the last prl_hip_reset_deskew_stats (process-wide: the chain searches on a helper thread).
static std::mutex g_deskew_stats_mu;
static prl_deskew_stats g_deskew_stats{};

// Segments of cv::HoughLinesP(~binarized, 1, CV_PI/180, threshold, line_length, line_gap) for `n_pages` 1-channel pages
// whose dark mask (p <= thr[page]) is the non-zero image.  `gray` pages are device resident.  Returns the segments
// per page in `lines_out` (host).  Synchronises the stream.  The caller holds ctx->ppht_mu (the workspace is ctx->ppht_buf).
static int ppht_pages(DeviceCtx* ctx, int n_pages, const PageSet& gray, int width, int height, int threshold, int line_length,
                      int line_gap, bool otsu, int fixed_thr, std::vector<std::vector<int>>* lines_out, std::vector<int>* thr_out,
                      hipStream_t stream, SearchStart* start = nullptr)
{
    const int numrho = (int)std::lrint((double)(float)((width + height) * 2 + 1));
    const size_t mask_page = r256((size_t)width * height);
    // fixed part of the workspace: hist, thr, row counts / offsets, counts, offsets, trig table
    const size_t b_hist = r256((size_t)n_pages * 256 * 4), b_thr = r256((size_t)n_pages * 4);
    const size_t b_rows = r256((size_t)n_pages * height * 4), b_cnt = r256((size_t)n_pages * 4), b_off = r256((size_t)n_pages * 8);
    const size_t b_ttab = r256(kNumAngle * 2 * 4);
    const size_t b_mask = mask_page * (size_t)n_pages;
    const size_t fixed = b_hist + b_thr + 2 * b_rows + 4 * b_cnt + 2 * b_off + b_ttab + b_mask;
    int st = ensure_buffer(&ctx->ppht_buf[0], &ctx->ppht_bytes[0], fixed);
    if (st != PRL_OK) return st;
    uint8_t* w = static_cast<uint8_t*>(ctx->ppht_buf[0]);
    auto take = [&](size_t bytes) { uint8_t* p = w; w += bytes; return p; };
    unsigned* d_hist = reinterpret_cast<unsigned*>(take(b_hist));
    int* d_thr = reinterpret_cast<int*>(take(b_thr));
    unsigned* d_rowcnt = reinterpret_cast<unsigned*>(take(b_rows));
    unsigned* d_rowoff = reinterpret_cast<unsigned*>(take(b_rows));
    unsigned* d_count = reinterpret_cast<unsigned*>(take(b_cnt));
    unsigned* d_nlines = reinterpret_cast<unsigned*>(take(b_cnt));
    unsigned* d_cap = reinterpret_cast<unsigned*>(take(b_cnt));
    unsigned long long* d_nzoff = reinterpret_cast<unsigned long long*>(take(b_off));
    unsigned long long* d_lnoff = reinterpret_cast<unsigned long long*>(take(b_off));
    float* d_ttab = reinterpret_cast<float*>(take(b_ttab));
    int* d_plist = reinterpret_cast<int*>(take(b_cnt));
    uint8_t* d_mask = take(b_mask);

    std::vector<int> h_thr((size_t)n_pages, fixed_thr);
    if (otsu) {
        PRL_HIP_CHECK(hipMemsetAsync(d_hist, 0, (size_t)n_pages * 256 * 4, stream));
        const dim3 hg((unsigned)std::min(4, (width + 1023) / 1024), (unsigned)std::min(height, 64), (unsigned)n_pages);
        hipLaunchKernelGGL(k_hist, hg, dim3(256), 0, stream, gray, width, height, d_hist);
        hipLaunchKernelGGL(k_otsu, dim3((unsigned)((n_pages + 63) / 64)), dim3(64), 0, stream, d_hist, width, height, n_pages, d_thr);
        PRL_HIP_CHECK(hipGetLastError());
    } else {
        PRL_HIP_CHECK(hipMemcpyAsync(d_thr, h_thr.data(), (size_t)n_pages * 4, hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(k_dark_mask, dim3((unsigned)height, (unsigned)n_pages), dim3(64), 0, stream, gray, width, height, d_thr, d_mask,
                       mask_page, d_rowcnt);
    hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)n_pages), dim3(256), 0, stream, height, d_rowcnt, d_rowoff, d_count);
    PRL_HIP_CHECK(hipGetLastError());
    std::vector<unsigned> h_count((size_t)n_pages);
    PRL_HIP_CHECK(hipMemcpyAsync(h_count.data(), d_count, (size_t)n_pages * 4, hipMemcpyDeviceToHost, stream));
    if (otsu) PRL_HIP_CHECK(hipMemcpyAsync(h_thr.data(), d_thr, (size_t)n_pages * 4, hipMemcpyDeviceToHost, stream));
    PRL_HIP_CHECK(hipStreamSynchronize(stream));
    if (thr_out) *thr_out = h_thr;

    // point lists and segment lists sized from the counts.  A good line clears at least line_length / (line_gap + 1)
    // points, which bounds the number of segments a page can produce.
    std::vector<unsigned long long> h_nzoff((size_t)n_pages), h_lnoff((size_t)n_pages);
    std::vector<unsigned> h_cap((size_t)n_pages);
    unsigned long long nz_total = 0, ln_total = 0;
    const unsigned per_line = (unsigned)std::max(1, line_length / (line_gap + 1));
    for (int i = 0; i < n_pages; ++i) {
        h_nzoff[(size_t)i] = nz_total;
        nz_total += (h_count[(size_t)i] + 63) / 64 * 64;
        h_cap[(size_t)i] = h_count[(size_t)i] / per_line + 16;
        h_lnoff[(size_t)i] = ln_total;
        ln_total += h_cap[(size_t)i];
    }
    const size_t b_nz = r256(nz_total * 4 + 256), b_lines = r256(ln_total * 16 + 256);
    // (growing synchronises the device, which would stall a chain that overlaps this search with its other stages:
    // keep a quarter of headroom, the lists of the following passes differ by the ink on their pages)
    if (ctx->ppht_bytes[1] < b_nz + b_lines) st = ensure_buffer(&ctx->ppht_buf[1], &ctx->ppht_bytes[1], (b_nz + b_lines) / 4 * 5);
    if (st != PRL_OK) return st;
    unsigned* d_nz = reinterpret_cast<unsigned*>(ctx->ppht_buf[1]);
    int* d_lines = reinterpret_cast<int*>(static_cast<uint8_t*>(ctx->ppht_buf[1]) + b_nz);
    PRL_HIP_CHECK(hipMemcpyAsync(d_nzoff, h_nzoff.data(), (size_t)n_pages * 8, hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_lnoff, h_lnoff.data(), (size_t)n_pages * 8, hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_cap, h_cap.data(), (size_t)n_pages * 4, hipMemcpyHostToDevice, stream));
    // the trig table of HoughLinesProbabilistic: (float)(cos((double)n * theta) * irho), host libm as in the reference
    float h_ttab[kNumAngle * 2];
    const float theta = (float)(3.1415926535897932384626433832795 / 180), irho = 1.f;
    for (int n = 0; n < kNumAngle; ++n) {
        h_ttab[2 * n] = (float)(std::cos((double)n * theta) * irho);
        h_ttab[2 * n + 1] = (float)(std::sin((double)n * theta) * irho);
    }
    PRL_HIP_CHECK(hipMemcpyAsync(d_ttab, h_ttab, sizeof(h_ttab), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_collect, dim3((unsigned)height, (unsigned)n_pages), dim3(64), 0, stream, width, height, d_mask, mask_page,
                       d_rowoff, d_nzoff, d_nz);
    PRL_HIP_CHECK(hipGetLastError());
    PRL_HIP_CHECK(hipMemsetAsync(d_nlines, 0, (size_t)n_pages * 4, stream));
    if (start && start->ev) PRL_HIP_CHECK(hipEventRecord(start->ev, stream));   // the streaming prelude ends here

    // ---- the transform.  Pages that qualify go to the group kernel (accumulator in LDS, ppht_group.hip); what it does not
    // finish - pages too large for int16 cells, a group that gave up waiting - is done by k_ppht_mw below, in this call. ----
    std::vector<unsigned> h_status((size_t)n_pages, 0u);
    std::vector<unsigned long long> h_gprof;
    PphtGroupIn gin;
    const int grp_env = env_knobs().ppht_group;
    bool group_ran = false;
    if (grp_env != 0 && !(start && start->prefer_mw && grp_env < 0) && ppht_group_eligible(width, height, threshold)) {
        gin.n_pages = n_pages; gin.width = width; gin.height = height; gin.threshold = threshold; gin.line_length = line_length;
        gin.line_gap = line_gap; gin.d_mask = d_mask; gin.mask_page = mask_page; gin.d_nz = d_nz; gin.d_nzoff = d_nzoff; gin.d_count = d_count;
        gin.d_ttab = d_ttab; gin.h_ttab = h_ttab; gin.d_lines = d_lines; gin.d_lnoff = d_lnoff; gin.d_cap = d_cap; gin.d_nlines = d_nlines;
        gin.h_count = h_count.data(); gin.h_nzoff = h_nzoff.data();
        gin.page_list.resize((size_t)n_pages);
        for (int i = 0; i < n_pages; ++i) gin.page_list[(size_t)i] = i;
        std::stable_sort(gin.page_list.begin(), gin.page_list.end(), [&](int x, int y) { return h_count[(size_t)x] > h_count[(size_t)y]; });
        gin.status_out = h_status.data();
#ifdef PRL_TEST_HOOKS
        if (std::getenv("PRL_HIP_PPHT_PROF")) { h_gprof.assign((size_t)n_pages * 16, 0ull); gin.prof_out = h_gprof.data(); }
#endif
        hipEvent_t dbg_ev[3] = {nullptr, nullptr, nullptr};
        if (env_knobs().debug) {
            for (hipEvent_t& e : dbg_ev) PRL_HIP_CHECK(hipEventCreate(&e));
            PRL_HIP_CHECK(hipEventRecord(dbg_ev[2], stream));
            gin.ev[0] = dbg_ev[0]; gin.ev[1] = dbg_ev[1];
        }
        struct EvFree { hipEvent_t* e; ~EvFree() { for (int i = 0; i < 3; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_free{dbg_ev};
        const int gst = ppht_group_run(ctx, gin, stream);
        if (gst == PRL_OK && env_knobs().debug) {
            PRL_HIP_CHECK(hipStreamSynchronize(stream));
            float ms_pre = 0, ms_k = 0;
            (void)hipEventElapsedTime(&ms_pre, dbg_ev[2], dbg_ev[0]);
            (void)hipEventElapsedTime(&ms_k, dbg_ev[0], dbg_ev[1]);
            std::fprintf(stderr, "[prl ppht] visiting order + bit masks %.2f ms, group kernel %.2f ms\n", ms_pre, ms_k);
        }
        group_ran = gst == PRL_OK;
        if (gst == PRL_ERR_NOMEM) return gst;
    }
    if (start) start->launched();
    std::vector<unsigned> h_nl((size_t)n_pages);
    std::vector<int> redo;
    if (group_ran) {
        PRL_HIP_CHECK(hipStreamSynchronize(stream));
        for (int i = 0; i < n_pages; ++i)
            if (h_status[(size_t)i] != 1u) redo.push_back(i);
        if (env_knobs().debug)
            std::fprintf(stderr, "[prl ppht] group kernel: %d pages, %d members per group, %d groups, %d workgroups, %d bytes of LDS; %zu pages left to k_ppht_mw\n",
                         n_pages, gin.geometry_out[0], gin.geometry_out[1], gin.geometry_out[2], gin.geometry_out[3], redo.size());
#ifdef PRL_TEST_HOOKS
        if (gin.prof_out) {
            std::fprintf(stderr, "{\"ppht_group_prof\": {\"pages\": %d, \"width\": %d, \"height\": %d, \"members\": %d, \"groups\": %d, \"per_page\": [", n_pages, width,
                         height, gin.geometry_out[0], gin.geometry_out[1]);
            std::vector<int> by_time(gin.page_list);
            auto cyc = [&](int pg) { unsigned long long t = 0; for (int k = 5; k < 12; ++k) t += h_gprof[(size_t)pg * 16 + k]; return t; };
            std::sort(by_time.begin(), by_time.end(), [&](int x, int y) { return cyc(x) > cyc(y); });
            for (int i = 0; i < std::min(n_pages, 4); ++i) {
                const int pg = by_time[(size_t)i];
                const unsigned long long* q = h_gprof.data() + (size_t)pg * 16;
                std::fprintf(stderr, "%s{\"page\": %d, \"points\": %u, \"exchanges\": %llu, \"blocks\": %llu, \"triggers\": %llu, \"good_lines\": %llu, \"walk_rounds\": %llu, \"cyc\": {\"fetch\": %llu, \"vote\": %llu, \"collect\": %llu, \"post\": %llu, \"walk\": %llu, \"erase\": %llu, \"strike\": %llu}, \"polls\": %llu, \"prefetched_polls\": %llu, \"collect_poll_cyc\": %llu, \"collect_barrier_cyc\": %llu}",
                             i ? ", " : "", pg, h_count[(size_t)pg], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12], q[13], q[14], q[15]);
            }
            std::fprintf(stderr, "], \"all_pages_points_triggers_mcycles\": [");
            for (int i = 0; i < n_pages; ++i) {
                const int pg = by_time[(size_t)i];
                std::fprintf(stderr, "%s[%u, %llu, %llu]", i ? ", " : "", h_count[(size_t)pg], h_gprof[(size_t)pg * 16 + 2], cyc(pg) / 1000000ull);
            }
            std::fprintf(stderr, "]}}\n");
        }
#endif
    } else {
        for (int i = 0; i < n_pages; ++i) redo.push_back(i);
    }
    PphtArgs a{};
    a.prof = nullptr;
#ifdef PRL_TEST_HOOKS
    unsigned long long* d_prof = nullptr;
    struct FreeProf { unsigned long long*& p; ~FreeProf() { if (p) (void)hipFree(p); } } free_prof{d_prof};
#endif
    if (!redo.empty()) {
        const int n_redo = (int)redo.size();
        const size_t b_accum = r256((size_t)n_redo * kNumAngle * numrho * 4);
        st = ensure_buffer(&ctx->ppht_buf[4], &ctx->ppht_bytes[4], b_accum);
        if (st != PRL_OK) return st;
        int* d_accum = static_cast<int*>(ctx->ppht_buf[4]);
        // (the paired un-votes address the cells as 64-bit words: every page's accumulator starts on an 8-byte boundary - hipMalloc
        // gives 256, a page is kNumAngle x numrho x 4 bytes with kNumAngle even - and a biased cell never borrows:
        // count > -2^31 + W H)
        if (reinterpret_cast<uintptr_t>(d_accum) % 8 != 0) return PRL_ERR_BAD_ARG;
        PRL_HIP_CHECK(hipMemcpyAsync(d_plist, redo.data(), (size_t)n_redo * 4, hipMemcpyHostToDevice, stream));
        PRL_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_accum), (int)kAccBias, (size_t)n_redo * kNumAngle * numrho, stream));   // count 0 = bias
        a.width = width; a.height = height; a.numrho = numrho; a.threshold = threshold; a.line_length = line_length; a.line_gap = line_gap;
        a.mask = d_mask; a.mask_page = mask_page; a.nz = d_nz; a.nz_off = d_nzoff; a.count = d_count; a.accum = d_accum; a.ttab = d_ttab;
        a.lines = d_lines; a.lines_off = d_lnoff; a.lines_cap = d_cap; a.n_lines = d_nlines;
        a.prio = env_knobs().ppht_prio;
        a.page_list = d_plist;
#ifdef PRL_TEST_HOOKS
        if (std::getenv("PRL_HIP_PPHT_PROF")) {
            PRL_HIP_CHECK(hipMalloc(&d_prof, (size_t)n_pages * 16 * 8));
            PRL_HIP_CHECK(hipMemsetAsync(d_prof, 0, (size_t)n_pages * 16 * 8, stream));
            a.prof = d_prof;
        }
#endif
        // three wavefronts per page (shorter critical path per page; equal to one per page once the memory system is the limit)
        const int mw_env = env_knobs().ppht_mw;
        if (mw_env == 1 || (mw_env < 0 && n_redo <= kPphtMwMaxPages)) hipLaunchKernelGGL(k_ppht_mw, dim3((unsigned)n_redo), dim3(64 * kMwWaves), 0, stream, a);
        else hipLaunchKernelGGL(k_ppht, dim3((unsigned)n_redo), dim3(64), 0, stream, a);
        PRL_HIP_CHECK(hipGetLastError());
    }
    // The segments: first how many each page has, then those and no more (the lists' capacity is a loose bound - a segment per
    // line_length / (line_gap + 1) points: 0.9 GB for 256 of the reference's scans, of which 7 MB hold segments).
    PRL_HIP_CHECK(hipMemcpyAsync(h_nl.data(), d_nlines, (size_t)n_pages * 4, hipMemcpyDeviceToHost, stream));
    PRL_HIP_CHECK(hipStreamSynchronize(stream));
    lines_out->assign((size_t)n_pages, {});
    for (int i = 0; i < n_pages; ++i) {
        const size_t have = std::min<size_t>(h_nl[(size_t)i], h_cap[(size_t)i]);
        (*lines_out)[(size_t)i].resize(have * 4);
        if (have)
            PRL_HIP_CHECK(hipMemcpyAsync((*lines_out)[(size_t)i].data(), d_lines + h_lnoff[(size_t)i] * 4, have * 16, hipMemcpyDeviceToHost, stream));
    }
    PRL_HIP_CHECK(hipStreamSynchronize(stream));
#ifdef PRL_TEST_HOOKS
    if (d_prof) {   // one JSON line per call on stderr: the three heaviest pages and the sum, cycles per phase and event counts
        std::vector<unsigned long long> hp((size_t)n_pages * 16);
        if (hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) std::fill(hp.begin(), hp.end(), 0ull);
        std::vector<int> order((size_t)n_pages);
        for (int i = 0; i < n_pages; ++i) order[(size_t)i] = i;
        auto total = [&](int i) { unsigned long long t = 0; for (int k = 0; k < 6; ++k) t += hp[(size_t)i * 16 + k]; return t; };
        std::sort(order.begin(), order.end(), [&](int x, int y) { return total(x) > total(y); });
        static const char* names[12] = {"prologue_cyc", "vote_cyc", "rollback_cyc", "walk1_cyc", "walk2_cyc", "unvote_cyc", "blocks", "triggers",
                                        "good_lines", "walk_chunks", "unvoted_px", "voted_points"};
        std::fprintf(stderr, "{\"ppht_prof\": {\"pages\": %d, \"width\": %d, \"height\": %d, \"heaviest\": [", n_pages, width, height);
        for (int r = 0; r < std::min(3, n_pages); ++r) {
            const int i = order[(size_t)r];
            std::fprintf(stderr, "%s{\"page\": %d, \"points\": %u, \"segments\": %u", r ? ", " : "", i, h_count[(size_t)i], h_nl[(size_t)i]);
            for (int k = 0; k < 12; ++k) std::fprintf(stderr, ", \"%s\": %llu", names[k], hp[(size_t)i * 16 + k]);
            std::fprintf(stderr, "}");
        }
        std::fprintf(stderr, "]}}\n");
    }
#endif
    {   // diagnostics of the search (prl_hip_last_deskew_stats): what the lists held against what they had room for
        std::lock_guard<std::mutex> lk(g_deskew_stats_mu);
        prl_deskew_stats& ds = g_deskew_stats;
        for (int i = 0; i < n_pages; ++i) {
            const long long room = (long long)h_cap[(size_t)i] - (long long)h_nl[(size_t)i];
            if (ds.pages == 0 || room < ds.min_page_headroom) ds.min_page_headroom = room;
            ds.pages += 1;
            ds.points += h_count[(size_t)i];
            ds.segments += h_nl[(size_t)i];
            ds.segment_capacity += h_cap[(size_t)i];
            ds.max_page_segments = std::max<uint64_t>(ds.max_page_segments, h_nl[(size_t)i]);
            ds.max_page_points = std::max<uint64_t>(ds.max_page_points, h_count[(size_t)i]);
        }
    }
    for (int i = 0; i < n_pages; ++i) {
        if (h_nl[(size_t)i] > h_cap[(size_t)i]) {
            set_error_detail("HoughLinesP: segment list overflow");
            return PRL_ERR_NOMEM;
        }
    }
    return PRL_OK;
}

static size_t ppht_bytes_per_page(int width, int height)
{
    const size_t numrho = (size_t)(width + height) * 2 + 1;
    return 2 * r256((size_t)width * height) + kNumAngle * numrho * 4 + (size_t)height * 8 + 2048 + (size_t)width * height / 2;
}


// pages per pass of prl::deskew: bounded by a workspace budget (mask + accumulator + point lists + gray page per page)
int deskew_pages_per_pass(int n_pages, int width, int height)
{
    const size_t budget = env_knobs().deskew_work_mb << 20;
    return (int)std::max<size_t>(1, std::min<size_t>({(size_t)n_pages, (size_t)16384, budget / ppht_bytes_per_page(width, height)}));
}

// Ink census (round 5): the number of points HoughLinesP will see on each page - the pixels at or below the page's Otsu threshold -
// from the gray histogram alone (one streaming read of the pages; no mask, no lists).  The chain sizes its passes with it: the
// search of a pass lasts as long as its heaviest page (0.6 us per point through one CU's atomic path), so a batch of photographs
// with a dark table in them wants few large passes, text scans the usual ones.  Synchronises `hs`.
int deskew_ink_census(DeviceCtx* ctx, int n_pages, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                      int height, std::vector<unsigned>* points, hipStream_t hs)
{
    points->assign((size_t)n_pages, 0u);
    const size_t gray_page = r256((size_t)width * height);
    const int sub = channels == 1 ? std::min(n_pages, 4096) : (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pages, ((size_t)512 << 20) / gray_page));
    std::lock_guard<std::mutex> lk(ctx->ppht_mu);
    int st = ensure_buffer(&ctx->ppht_buf[0], &ctx->ppht_bytes[0], r256((size_t)sub * 256 * 4) + r256((size_t)sub * 4));
    if (st != PRL_OK) return st;
    unsigned* d_hist = static_cast<unsigned*>(ctx->ppht_buf[0]);
    int* d_thr = reinterpret_cast<int*>(static_cast<uint8_t*>(ctx->ppht_buf[0]) + r256((size_t)sub * 256 * 4));
    if (channels != 1) {
        st = ensure_buffer(&ctx->ppht_buf[2], &ctx->ppht_bytes[2], gray_page * (size_t)sub);
        if (st != PRL_OK) return st;
    }
    std::vector<unsigned> h_hist((size_t)sub * 256);
    std::vector<int> h_thr((size_t)sub);
    for (int first = 0; first < n_pages; first += sub) {
        const int cnt = std::min(sub, n_pages - first);
        PageSet g{};
        if (channels != 1) {
            uint8_t* gray_ws = static_cast<uint8_t*>(ctx->ppht_buf[2]);
            st = prl_hip_bgr2gray_batch_device(cnt, channels, src + (size_t)first * src_page_stride, src_page_stride, src_step, width, height,
                                               gray_ws, gray_page, (size_t)width, hs);
            if (st != PRL_OK) return st;
            g.base = gray_ws; g.page_stride = gray_page; g.step = (size_t)width;
        } else {
            g.base = src + (size_t)first * src_page_stride; g.page_stride = src_page_stride; g.step = src_step;
        }
        PRL_HIP_CHECK(hipMemsetAsync(d_hist, 0, (size_t)cnt * 256 * 4, hs));
        const dim3 hg((unsigned)std::min(4, (width + 1023) / 1024), (unsigned)std::min(height, 64), (unsigned)cnt);
        hipLaunchKernelGGL(k_hist, hg, dim3(256), 0, hs, g, width, height, d_hist);
        hipLaunchKernelGGL(k_otsu, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, hs, d_hist, width, height, cnt, d_thr);
        PRL_HIP_CHECK(hipGetLastError());
        PRL_HIP_CHECK(hipMemcpyAsync(h_hist.data(), d_hist, (size_t)cnt * 256 * 4, hipMemcpyDeviceToHost, hs));
        PRL_HIP_CHECK(hipMemcpyAsync(h_thr.data(), d_thr, (size_t)cnt * 4, hipMemcpyDeviceToHost, hs));
        PRL_HIP_CHECK(hipStreamSynchronize(hs));
        for (int i = 0; i < cnt; ++i) {
            unsigned long long n = 0;
            for (int v = 0; v <= std::min(255, std::max(-1, h_thr[(size_t)i])); ++v) n += h_hist[(size_t)i * 256 + v];
            (*points)[(size_t)(first + i)] = (unsigned)std::min<unsigned long long>(n, 0xffffffffull);
        }
    }
    return PRL_OK;
}

// First half of prl::deskew on `cnt` pages (cnt <= deskew_pages_per_pass): gray -> Otsu -> HoughLinesP -> angle vote.
int deskew_find(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                int height, DeskewPlan* plan, hipStream_t hs, SearchStart* start)
{
    const size_t gray_page = r256((size_t)width * height);
    std::lock_guard<std::mutex> lk(ctx->ppht_mu);
    int st;
    PageSet g{};
    if (channels != 1) {  // deskew.cpp:214-217
        st = ensure_buffer(&ctx->ppht_buf[2], &ctx->ppht_bytes[2], gray_page * (size_t)cnt);
        if (st != PRL_OK) return st;
        uint8_t* gray_ws = static_cast<uint8_t*>(ctx->ppht_buf[2]);
        st = prl_hip_bgr2gray_batch_device(cnt, channels, src, src_page_stride, src_step, width, height, gray_ws, gray_page,
                                           (size_t)width, hs);
        if (st != PRL_OK) return st;
        g.base = gray_ws; g.page_stride = gray_page; g.step = (size_t)width;
    } else {
        g.base = src; g.page_stride = src_page_stride; g.step = src_step;
    }
    std::vector<std::vector<int>> lines;
    // cv::threshold(..., THRESH_BINARY | THRESH_OTSU) (:224) + findAngle's bitwise_not (:146): points = (p <= otsu)
    st = ppht_pages(ctx, cnt, g, width, height, 100, (int)std::lrint((double)(width / 8.f)), (int)std::lrint(20.0), true, 0, &lines,
                    nullptr, hs, start);
    if (st != PRL_OK) return st;
    plan->warp.resize(sizeof(WarpPage) * (size_t)cnt);
    plan->wh.resize(2 * (size_t)cnt);
    plan->angles.resize((size_t)cnt);
    plan->max_ow = plan->max_oh = 0;
    WarpPage* wp = reinterpret_cast<WarpPage*>(plan->warp.data());
    for (int i = 0; i < cnt; ++i) {
        const double angle = vote_angle(lines[(size_t)i].data(), (int)(lines[(size_t)i].size() / 4));
        plan->angles[(size_t)i] = angle;
        const bool rot = (angle != 0) && (angle <= DBL_MAX && angle >= -DBL_MAX);  // deskew.cpp:228
        fill_warp_page(width, height, angle, !rot, &wp[i]);
        plan->wh[2 * (size_t)i] = wp[i].ow;
        plan->wh[2 * (size_t)i + 1] = wp[i].oh;
        plan->max_ow = std::max(plan->max_ow, wp[i].ow);
        plan->max_oh = std::max(plan->max_oh, wp[i].oh);
    }
    return PRL_OK;
}

// Second half: rotate.
int deskew_apply(DeviceCtx* ctx, const DeskewPlan& plan, int cnt, int channels, const uint8_t* src, size_t src_page_stride,
                 size_t src_step, int width, int height, uint8_t* dst, size_t dst_page_stride, size_t dst_step, hipStream_t hs)
{
    if (plan.warp.size() != sizeof(WarpPage) * (size_t)cnt) return PRL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(hs, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    int st = ensure_small(ctx, plan.warp.size());
    if (st != PRL_OK) return st;
    ctx->lut_small[0] = ctx->lut_small[1] = nullptr;
    PRL_HIP_CHECK(hipMemcpyAsync(ctx->small, plan.warp.data(), plan.warp.size(), hipMemcpyHostToDevice, hs));
    PRL_HIP_CHECK(hipStreamSynchronize(hs));  // pageable source
    PageSet s{};
    s.base = src; s.page_stride = src_page_stride; s.step = src_step;
    PageSetOut d{};
    d.base = dst; d.page_stride = dst_page_stride; d.step = dst_step;
    st = launch_warp(channels, s, d, width, height, cnt, plan.max_ow, plan.max_oh, static_cast<const WarpPage*>(ctx->small), hs);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, hs));
    return PRL_OK;
}

int deskew_pages(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                 int height, uint8_t* dst, size_t dst_page_stride, size_t dst_step, int32_t* out_wh, double* angles, hipStream_t hs)
{
    DeskewPlan plan;
    int st = deskew_find(ctx, cnt, channels, src, src_page_stride, src_step, width, height, &plan, hs);
    if (st != PRL_OK) return st;
    std::copy(plan.wh.begin(), plan.wh.end(), out_wh);
    if (angles) std::copy(plan.angles.begin(), plan.angles.end(), angles);
    return deskew_apply(ctx, plan, cnt, channels, src, src_page_stride, src_step, width, height, dst, dst_page_stride, dst_step, hs);
}

}  // namespace prl_hip

using namespace prl_hip;

namespace {
// one page host -> device -> host around a device-side stage that produces a page of per-page size
template <typename F>
int host_roundtrip(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst, size_t dst_step,
                   int max_w, int max_h, int* out_w, int* out_h, F&& stage)
{
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    const size_t in_row = (size_t)width * channels, out_row_max = (size_t)max_w * channels;
    const size_t in_bytes = r256(in_row * (size_t)height), out_bytes = r256(out_row_max * (size_t)max_h);
    uint8_t *d_in = nullptr, *d_out = nullptr;
    // own buffers: the stage itself uses the shared staging area for its gray pages
    PRL_HIP_CHECK(hipMalloc(&d_in, in_bytes + 2 * out_bytes));
    struct Free { uint8_t* p; ~Free() { (void)hipFree(p); } } free_d_in{d_in};  // every exit, the error returns of PRL_HIP_CHECK included
    d_out = d_in + in_bytes;
    uint8_t* d_packed = d_out + out_bytes;  // the result with rows packed to its own width
    hipStream_t stream = nullptr;
    {
        std::lock_guard<std::mutex> slk(ctx->stage_mu);
        st = ensure_stage_pinned(ctx, std::max(in_bytes, out_bytes));
        if (st == PRL_OK) st = stage_upload(ctx, 0, src, src_step, in_row, height, d_in, stream);
        if (st == PRL_OK) st = hipStreamSynchronize(stream) == hipSuccess ? PRL_OK : PRL_ERR_HIP;
    }
    if (st == PRL_OK) st = stage(d_in, in_bytes, in_row, d_out, out_bytes, out_row_max, stream);
    if (st == PRL_OK && (dst_step < (size_t)*out_w * channels)) st = PRL_ERR_BAD_ARG;
    if (st == PRL_OK) {
        std::lock_guard<std::mutex> slk(ctx->stage_mu);
        // rows of the result are out_row_max apart on the device: fetch them packed
        const size_t out_row = (size_t)*out_w * channels;
        PRL_HIP_CHECK(hipMemcpy2DAsync(d_packed, out_row, d_out, out_row_max, out_row, (size_t)*out_h, hipMemcpyDeviceToDevice, stream));
        st = stage_download(ctx, 0, d_packed, out_row, *out_h, dst, dst_step, stream);
    }
    return st;
}
}  // namespace

extern "C" {

int prl_hip_rotate_out_size(int width, int height, double angle, int* out_w, int* out_h)
{
    if (!out_w || !out_h) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    WarpPage wp{};
    fill_warp_page(width, height, angle, false, &wp);
    *out_w = wp.ow;
    *out_h = wp.oh;
    return PRL_OK;
}

// prl::rotate on device pages, one angle per page (angles == NULL: not allowed).
int prl_hip_rotate_batch_device(int n_pages, int channels, const double* angles, const uint8_t* d_src, size_t src_page_stride,
                                size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels < 1 || channels > 4) return PRL_ERR_BAD_CHANNELS;
    if (n_pages < 0 || !angles || !d_src || !d_dst || d_src == d_dst || src_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    if (std::max(width, height) > 32767) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int chunk = std::min(n_pages, 32768);
    st = ensure_small(ctx, sizeof(WarpPage) * (size_t)chunk);
    if (st != PRL_OK) return st;
    ctx->lut_small[0] = ctx->lut_small[1] = nullptr;
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(hs, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    std::vector<WarpPage> wp((size_t)chunk);
    for (int first = 0; first < n_pages; first += chunk) {
        const int cnt = std::min(chunk, n_pages - first);
        int max_ow = 0, max_oh = 0;
        for (int i = 0; i < cnt; ++i) {
            fill_warp_page(width, height, angles[first + i], false, &wp[(size_t)i]);
            max_ow = std::max(max_ow, wp[(size_t)i].ow);
            max_oh = std::max(max_oh, wp[(size_t)i].oh);
            if (dst_step < (size_t)wp[(size_t)i].ow * channels) return PRL_ERR_BAD_ARG;
        }
        PRL_HIP_CHECK(hipMemcpyAsync(ctx->small, wp.data(), sizeof(WarpPage) * (size_t)cnt, hipMemcpyHostToDevice, hs));
        PRL_HIP_CHECK(hipStreamSynchronize(hs));  // `wp` is pageable host memory reused by the next chunk
        PageSet s{};
        s.base = d_src + (size_t)first * src_page_stride; s.page_stride = src_page_stride; s.step = src_step;
        PageSetOut d{};
        d.base = d_dst + (size_t)first * dst_page_stride; d.page_stride = dst_page_stride; d.step = dst_step;
        st = launch_warp(channels, s, d, width, height, cnt, max_ow, max_oh, static_cast<const WarpPage*>(ctx->small), hs);
        if (st != PRL_OK) return st;
    }
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, hs));
    return PRL_OK;
}

// cv::HoughLinesP(image, lines, 1, CV_PI/180, threshold, line_length, line_gap) on ONE 1-channel device page whose
// non-zero pixels are the points (test / building-block entry).  lines: host buffer of 4*cap ints; *n_lines = segments found.
int prl_hip_houghp_device(const uint8_t* d_image, size_t step, int width, int height, int threshold, int line_length, int line_gap,
                          int32_t* lines, int cap, int* n_lines, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (!d_image || !n_lines || (cap > 0 && !lines) || step < (size_t)width || std::max(width, height) > 32767) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> slk(ctx->stage_mu);  // lock order: stage_mu, then mu
    // HoughLinesP's points are the NON-ZERO pixels; the point stage selects "p <= thr", so it runs on the complement:
    // p != 0  <=>  (255 - p) <= 254
    const size_t bytes = (size_t)width * height;
    st = ensure_stage(ctx, bytes);
    if (st != PRL_OK) return st;
    st = stage_acquire(ctx, hs);
    if (st != PRL_OK) return st;
    StageRelease release{ctx, hs};
    st = prl_hip_invert_batch_device(1, d_image, 0, step, width, height, static_cast<uint8_t*>(ctx->stage), bytes, (size_t)width, stream);
    if (st != PRL_OK) return st;
    PageSet g{};
    g.base = static_cast<uint8_t*>(ctx->stage); g.page_stride = 0; g.step = (size_t)width;
    std::vector<std::vector<int>> out;
    {
        std::lock_guard<std::mutex> lk(ctx->ppht_mu);
        st = ppht_pages(ctx, 1, g, width, height, threshold, line_length, line_gap, false, 254, &out, nullptr, hs);
    }
    if (st != PRL_OK) return st;
    *n_lines = (int)(out[0].size() / 4);
    for (int i = 0; i < std::min(cap, *n_lines) * 4; ++i) lines[i] = out[0][(size_t)i];
    return PRL_OK;
}

int prl_hip_last_deskew_stats(prl_deskew_stats* out)
{
    if (!out) return PRL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_deskew_stats_mu);
    *out = g_deskew_stats;
    return PRL_OK;
}

int prl_hip_reset_deskew_stats(void)
{
    std::lock_guard<std::mutex> lk(g_deskew_stats_mu);
    g_deskew_stats = prl_deskew_stats{};
    return PRL_OK;
}

/*
 * prl::findAngle (src/deskew/deskew.cpp:139-205) on n_pages 1-channel device pages: bitwise_not (:146), cv::HoughLinesP(input,
 * lines, 1, CV_PI/180, 100, width/8.f, 20) (:148), atan2 per segment and the first-fit vote (:158-201).  HoughLinesP's points
 * are the non-zero pixels of the complement, i.e. the pixels p != 255: the point stage runs on the page as it is with the
 * fixed threshold 254 (no inverted copy).  angles / n_segments: host arrays, one entry per page.  Synchronises.
 */
int prl_hip_find_angle_batch_device(int n_pages, const uint8_t* d_image, size_t page_stride, size_t step, int width, int height,
                                    double* angles, int32_t* n_segments, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (n_pages < 0 || !d_image || !angles || step < (size_t)width || std::max(width, height) > 32767) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const int chunk = deskew_pages_per_pass(n_pages, width, height);
    for (int first = 0; first < n_pages; first += chunk) {
        const int cnt = std::min(chunk, n_pages - first);
        PageSet g{};
        g.base = d_image + (size_t)first * page_stride; g.page_stride = page_stride; g.step = step;
        std::vector<std::vector<int>> lines;
        {
            std::lock_guard<std::mutex> lk(ctx->ppht_mu);
            st = ppht_pages(ctx, cnt, g, width, height, 100, (int)std::lrint((double)(width / 8.f)), (int)std::lrint(20.0), false, 254,
                            &lines, nullptr, hs);
        }
        if (st != PRL_OK) return st;
        for (int i = 0; i < cnt; ++i) {
            const int n = (int)(lines[(size_t)i].size() / 4);
            angles[first + i] = vote_angle(lines[(size_t)i].data(), n);
            if (n_segments) n_segments[first + i] = n;
        }
    }
    return PRL_OK;
}

/* prl::findAngle on one host image (what the cv::Mat wrapper calls). */
int prl_hip_find_angle_host(const uint8_t* src, size_t src_step, int width, int height, double* angle, int32_t* n_segments)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY;
    if (!angle || src_step < (size_t)width || std::max(width, height) > 32767) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    const size_t bytes = r256((size_t)width * height);
    uint8_t* d_in = nullptr;
    PRL_HIP_CHECK(hipMalloc(&d_in, bytes));
    struct Free { uint8_t* p; ~Free() { (void)hipFree(p); } } free_d_in{d_in};
    hipStream_t stream = nullptr;
    {
        std::lock_guard<std::mutex> slk(ctx->stage_mu);
        st = ensure_stage_pinned(ctx, bytes);
        if (st == PRL_OK) st = stage_upload(ctx, 0, src, src_step, (size_t)width, height, d_in, stream);
        if (st == PRL_OK) st = hipStreamSynchronize(stream) == hipSuccess ? PRL_OK : PRL_ERR_HIP;
    }
    if (st != PRL_OK) return st;
    return prl_hip_find_angle_batch_device(1, d_in, bytes, (size_t)width, width, height, angle, n_segments, stream);
}

/*
 * prl::deskew on n_pages device pages (channels 1, 3 or 4).  Every page's result has its own size: len x len
 * (len = max(width, height)) when an angle was found, width x height otherwise (and transposed / same size for exactly
 * +-90 / 180 degrees); out_wh (host, 2 ints per page) receives it, angles (host, optional) findAngle's result in degrees.
 * d_dst pages must have room for len rows of dst_step >= len * channels bytes.  Synchronises the stream.
 */
int prl_hip_deskew_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                int32_t* out_wh, double* angles, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;  // CV_Assert(!inputImage.empty()), deskew.cpp:210
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    const int len = std::max(width, height);
    if (n_pages < 0 || !d_src || !d_dst || d_src == d_dst || !out_wh || src_step < (size_t)width * channels ||
        dst_step < (size_t)len * channels || len > 32767)
        return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    const int chunk = deskew_pages_per_pass(n_pages, width, height);
    for (int first = 0; first < n_pages; first += chunk) {
        const int cnt = std::min(chunk, n_pages - first);
        st = deskew_pages(ctx, cnt, channels, d_src + (size_t)first * src_page_stride, src_page_stride, src_step, width, height,
                          d_dst + (size_t)first * dst_page_stride, dst_page_stride, dst_step, out_wh + 2 * first,
                          angles ? angles + first : nullptr, static_cast<hipStream_t>(stream));
        if (st != PRL_OK) return st;
    }
    return PRL_OK;
}

/* prl::rotate on one host image; dst must hold the size prl_hip_rotate_out_size reports. */
int prl_hip_rotate_host(int channels, double angle, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                        size_t dst_step)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY;
    if (channels < 1 || channels > 4) return PRL_ERR_BAD_CHANNELS;
    if (!dst || src_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    int ow = 0, oh = 0;
    int st = prl_hip_rotate_out_size(width, height, angle, &ow, &oh);
    if (st != PRL_OK) return st;
    return host_roundtrip(channels, src, src_step, width, height, dst, dst_step, ow, oh, &ow, &oh,
                          [&](uint8_t* d_in, size_t in_bytes, size_t in_row, uint8_t* d_out, size_t out_bytes, size_t out_row, hipStream_t s) {
                              return prl_hip_rotate_batch_device(1, channels, &angle, d_in, in_bytes, in_row, width, height, d_out,
                                                                 out_bytes, out_row, s);
                          });
}

/* prl::deskew on one host image; dst must have room for max(width,height)^2 pixels (dst_step >= that side * channels);
 * *out_w x *out_h is the size of the result, *angle (optional) findAngle's degrees. */
int prl_hip_deskew_host(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst, size_t dst_step,
                        int* out_w, int* out_h, double* angle)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY;
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (!dst || !out_w || !out_h || src_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    const int len = std::max(width, height);
    int32_t wh[2] = {0, 0};
    double ang = 0;
    const int st = host_roundtrip(channels, src, src_step, width, height, dst, dst_step, len, len, &wh[0], &wh[1],
                                  [&](uint8_t* d_in, size_t in_bytes, size_t in_row, uint8_t* d_out, size_t out_bytes, size_t out_row, hipStream_t s) {
                                      return prl_hip_deskew_batch_device(1, channels, d_in, in_bytes, in_row, width, height, d_out,
                                                                         out_bytes, out_row, wh, &ang, s);
                                  });
    *out_w = wh[0];
    *out_h = wh[1];
    if (angle) *angle = ang;
    return st;
}

}  // extern "C"
