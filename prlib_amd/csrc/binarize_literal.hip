// binarize_literal.hip — the reference-shaped pipeline on the GPU: materialised float64 integral
// images, then one literal evaluation per output pixel.
//
// It reproduces, stage by stage, what prl::binarize* asks OpenCV to do
// (src/binarizations/binarizeSauvola.cpp:65-122 and the same lines of the other four files):
//   k_row_prefix   cv::copyMakeBorder(REPLICATE) [never materialised: clamped reads] + the row half
//                  of cv::integral — one wavefront per padded row, wave-level prefix scan
//   k_col_accum    the column half of cv::integral; values become the float64 the reference stores
//   k_dev_max      Wolf-Jolion's cv::minMaxLoc(localDevianceValues)
//   k_threshold    filter2D x2, mul, -=, sqrt, threshold formula, convertTo(CV_8U), compare
//
// This path moves ~50 B per pixel through HBM and is not the fast path; it is (1) the executable
// specification the fused kernel is validated against on the device and (2) the fallback for pages
// whose fix-up list overflows.  Roofline: HBM-bound, algorithmic 2 B/px, actual ~48 B/px.
#include "prl_device_math.h"
#include "prl_internal.h"

namespace prl_hip {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// inclusive prefix sum across the 64 lanes of a wavefront (Hillis-Steele over __shfl_up)
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v)
{
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned t = __shfl_up(v, d, kWave);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ void k_init_globals(PageGlobals* g, int n, unsigned* counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (counters)   // the fused pipeline's counter block (one launch instead of two)
        for (int j = i; j < kFusedCounterWords; j += gridDim.x * blockDim.x) counters[j] = 0u;
    if (i < n) {
        g[i].imin = 255;
        g[i].smax_found = 0;
        g[i].smax_bits = 0ull;
        g[i].coeff = 0.0;
        g[i].n_refined = 0;
        g[i].n_exact = 0;
        g[i].worklist_overflow = 0;
        g[i].v32max_bits = 0;
        g[i].n_cand = 0;
        g[i].need_literal = 0;
        g[i].kmax_bits = 0ull;
        g[i].coeff_rel = 0.0;
        g[i].cand_overflow = 0;
        g[i].reserved0 = 0;
    }
}

// Page minimum (cv::minMaxLoc(imageInput, &imageMin), binarizeWolfJolion.cpp:115-116).  The padded
// image has the same minimum as the page, so the page itself is reduced.  One wavefront per row chunk,
// 16 B per lane per load (rows are walked with their own alignment; head/tail bytes one by one).
__device__ __forceinline__ unsigned min4(unsigned mn, unsigned w)
{
    mn = min(mn, w & 0xffu);
    mn = min(mn, (w >> 8) & 0xffu);
    mn = min(mn, (w >> 16) & 0xffu);
    return min(mn, w >> 24);
}

__global__ void __launch_bounds__(256) k_page_min(PageSet src, int width, int height, PageGlobals* g)
{
    const int page = blockIdx.y;
    const uint8_t* img = src.page(page);
    unsigned mn = 255;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, n_waves = gridDim.x * blockDim.x / kWave;
    for (int y = wave; y < height; y += n_waves) {
        const uint8_t* row = img + (size_t)y * src.step;
        const int head = (int)((16 - ((size_t)row & 15)) & 15);
        const int nvec = (width - min(head, width)) / 16;
        for (int x = lane; x < min(head, width); x += kWave) mn = min(mn, (unsigned)row[x]);
        const uint4* v = reinterpret_cast<const uint4*>(row + head);
        for (int i = lane; i < nvec; i += kWave) {
            const uint4 q = v[i];
            mn = min4(min4(min4(min4(mn, q.x), q.y), q.z), q.w);
        }
        for (int x = head + nvec * 16 + lane; x < width; x += kWave) mn = min(mn, (unsigned)row[x]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned o = __shfl_xor(mn, d, kWave);
        mn = o < mn ? o : mn;
    }
    if (lane == 0) atomicMin(&g[page].imin, (int)mn);
}

// Row half of cv::integral over the replicate-padded page.  One wavefront per padded row; each lane
// owns 4 consecutive columns per step; lane totals go through a wave-level prefix scan and a scalar
// carry walks along the row.  Output: inclusive row prefixes of P and P*P as u64.
__global__ void __launch_bounds__(256) k_row_prefix(PageSet src, ThrParams tp, int first_page,
                                                   unsigned long long* __restrict__ ii,
                                                   unsigned long long* __restrict__ iq, size_t plane)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int page = blockIdx.y;
    if (row >= tp.ph) return;
    const uint8_t* img = src.page(first_page + page);
    const int sy = clampi(row - tp.half, 0, tp.height - 1);
    const uint8_t* srow = img + (size_t)sy * src.step;
    unsigned long long* orow_s = ii + (size_t)page * plane + (size_t)row * tp.pw;
    unsigned long long* orow_q = iq + (size_t)page * plane + (size_t)row * tp.pw;

    unsigned long long carry_s = 0, carry_q = 0;
    for (int x0 = 0; x0 < tp.pw; x0 += kWave * 4) {
        const int xb = x0 + lane * 4;
        unsigned ps[4], pq[4];
        unsigned accs = 0, accq = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = xb + c;
            unsigned v = 0;
            if (x < tp.pw) v = srow[clampi(x - tp.half, 0, tp.width - 1)];
            accs += v;
            accq += v * v;
            ps[c] = accs;
            pq[c] = accq;
        }
        const unsigned incs = wave_inclusive_scan(accs);
        const unsigned incq = wave_inclusive_scan(accq);
        const unsigned long long base_s = carry_s + (incs - accs);
        const unsigned long long base_q = carry_q + (incq - accq);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = xb + c;
            if (x < tp.pw) {
                orow_s[x] = base_s + ps[c];
                orow_q[x] = base_q + pq[c];
            }
        }
        carry_s += __shfl(incs, kWave - 1, kWave);
        carry_q += __shfl(incq, kWave - 1, kWave);
    }
}

// Column half of cv::integral: one thread per padded column walks down the rows.  The u64 row
// prefixes are replaced in place by the float64 values cv::integral(..., CV_64F) stores; every value
// is an integer < 2^53, so the conversion is exact (binarizeSauvola.cpp:72).
__global__ void __launch_bounds__(256) k_col_accum(ThrParams tp, unsigned long long* __restrict__ ii,
                                                  unsigned long long* __restrict__ iq, size_t plane)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int page = blockIdx.y;
    if (x >= tp.pw) return;
    unsigned long long* ps = ii + (size_t)page * plane + x;
    unsigned long long* pq = iq + (size_t)page * plane + x;
    unsigned long long as = 0, aq = 0;
    for (int y = 0; y < tp.ph; ++y) {
        as += ps[(size_t)y * tp.pw];
        aq += pq[(size_t)y * tp.pw];
        reinterpret_cast<double*>(ps)[(size_t)y * tp.pw] = (double)as;
        reinterpret_cast<double*>(pq)[(size_t)y * tp.pw] = (double)aq;
    }
}

__device__ __forceinline__ void mean_dev_at(const ThrParams& tp, const double* __restrict__ ii,
                                            const double* __restrict__ iq, int y, int x, double* m,
                                            double* s)
{
    const size_t r0 = (size_t)y * tp.pw + x;
    const size_t r1 = (size_t)(y + tp.w - 1) * tp.pw + x;
    const int o = tp.w - 1;
    const double mm = box4_literal(ii[r0], ii[r0 + o], ii[r1], ii[r1 + o], tp.f);
    const double q = box4_literal(iq[r0], iq[r0 + o], iq[r1], iq[r1 + o], tp.f);
    *m = mm;
    *s = dev_from(mm, q);
}

// Wolf-Jolion: cv::minMaxLoc(localDevianceValues, &min, &max)  — binarizeWolfJolion.cpp:118-119.
// [upstream] NaN never wins a comparison; deviations are >= +0, so the u64 bit patterns order
// exactly like the values and atomicMax on the bits is an exact float64 max.
__global__ void __launch_bounds__(256) k_dev_max(ThrParams tp, const double* __restrict__ ii,
                                                const double* __restrict__ iq, size_t plane,
                                                int first_page, PageGlobals* g)
{
    const int page = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    unsigned long long bits = 0;
    int found = 0;
    if (x < tp.ow) {
        double m, s;
        mean_dev_at(tp, ii + (size_t)page * plane, iq + (size_t)page * plane, y, x, &m, &s);
        if (s == s) {
            found = 1;
            bits = (unsigned long long)__double_as_longlong(s) & 0x7fffffffffffffffull;  // -0 -> +0
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(bits, d, kWave);
        bits = o > bits ? o : bits;
        found |= __shfl_xor(found, d, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && found) {
        atomicMax(&g[first_page + page].smax_bits, bits);
        atomicOr(&g[first_page + page].smax_found, 1);
    }
}

// double coeff = k / devianceMax  — binarizeWolfJolion.cpp:121 (IEEE division on the device)
__global__ void k_wolf_coeff(ThrParams tp, PageGlobals* g, int first_page, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PageGlobals& pg = g[first_page + i];
    const double smax = pg.smax_found ? __longlong_as_double((long long)pg.smax_bits)
                                      : -1.7976931348623157e308;  // minMaxLoc's initial -DBL_MAX
    pg.coeff = tp.k / smax;
}

__global__ void __launch_bounds__(256) k_threshold(PageSet src, ThrParams tp, int first_page,
                                                  const double* __restrict__ ii,
                                                  const double* __restrict__ iq, size_t plane,
                                                  PageSetOut dst, const PageGlobals* __restrict__ g)
{
    const int page = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= tp.ow) return;
    double m, s;
    mean_dev_at(tp, ii + (size_t)page * plane, iq + (size_t)page * plane, y, x, &m, &s);
    const PageGlobals& pg = g[first_page + page];
    const double T = threshold_literal(tp, m, s, (double)pg.imin, pg.coeff);
    // imageInput(processingRect): padded(half+y, half+x) == page(y, x) for every output position
    const unsigned p = src.page(first_page + page)[(size_t)y * src.step + x];
    dst.page(first_page + page)[(size_t)y * dst.step + x] = decide_literal(p, T);
}

}  // namespace

int wolf_coeff_run(const ThrParams& tp, PageGlobals* d_globals, int first_page, int n_pages, hipStream_t stream)
{
    hipLaunchKernelGGL(k_wolf_coeff, dim3((n_pages + 63) / 64), dim3(64), 0, stream, tp, d_globals, first_page,
                       n_pages);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

size_t literal_scratch_per_page(const ThrParams& tp)
{
    return 2 * sizeof(double) * (size_t)tp.pw * (size_t)tp.ph;
}

int init_globals_run(PageGlobals* d_globals, int n_pages, hipStream_t stream, void* fused_counters)
{
    hipLaunchKernelGGL(k_init_globals, dim3((n_pages + 255) / 256), dim3(256), 0, stream, d_globals,
                       n_pages, static_cast<unsigned*>(fused_counters));
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

int page_min_run(const ThrParams& tp, const PageSet& src, int n_pages, PageGlobals* d_globals,
                 hipStream_t stream)
{
    int blocks = (tp.height + 15) / 16;  // 4 wavefronts per block, ~4 rows each
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_page_min, dim3(blocks, n_pages), dim3(256), 0, stream, src, tp.width,
                       tp.height, d_globals);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// Runs pages [first_page, first_page + n_pages) of the batch; `scratch` must hold
// n_pages * literal_scratch_per_page(tp) bytes.  Globals must already hold imin where needed.
int literal_run(const ThrParams& tp, const PageSet& src, int first_page, int n_pages,
                const PageSetOut& dst, void* scratch, PageGlobals* d_globals, hipStream_t stream)
{
    const size_t plane = (size_t)tp.pw * tp.ph;
    auto* ii = reinterpret_cast<unsigned long long*>(scratch);
    auto* iq = ii + plane * (size_t)n_pages;

    const int rows_per_block = 256 / kWave;
    hipLaunchKernelGGL(k_row_prefix, dim3((tp.ph + rows_per_block - 1) / rows_per_block, n_pages),
                       dim3(256), 0, stream, src, tp, first_page, ii, iq, plane);
    PRL_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_col_accum, dim3((tp.pw + 255) / 256, n_pages), dim3(256), 0, stream, tp, ii, iq,
                       plane);
    PRL_HIP_CHECK(hipGetLastError());

    const auto* dii = reinterpret_cast<const double*>(ii);
    const auto* diq = reinterpret_cast<const double*>(iq);
    const dim3 grid((tp.ow + 255) / 256, tp.oh, n_pages);
    if (tp.method == PRL_WOLFJOLION) {
        hipLaunchKernelGGL(k_dev_max, grid, dim3(256), 0, stream, tp, dii, diq, plane, first_page,
                           d_globals);
        PRL_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(k_wolf_coeff, dim3((n_pages + 63) / 64), dim3(64), 0, stream, tp, d_globals,
                           first_page, n_pages);
        PRL_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_threshold, grid, dim3(256), 0, stream, src, tp, first_page, dii, diq, plane, dst,
                       d_globals);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

}  // namespace prl_hip
