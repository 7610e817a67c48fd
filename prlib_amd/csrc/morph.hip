// morph.hip — the cv::dilate / cv::erode pair at the end of every binarizer
// (src/binarizations/binarizeSauvola.cpp:125-134):
//   n > 0 : dilate(n) then erode(n)   (closing)
//   n < 0 : erode(|n|) then dilate(|n|) (opening)
// [upstream] cv::dilate(src, dst, Mat(), Point(-1,-1), n) = 3x3 rectangle iterated n times, which
// OpenCV folds into one (2n+1)x(2n+1) rectangle; the default border value makes out-of-image pixels
// neutral (ignored) for both operations.
//
// One workgroup produces a 64x32 output tile.  The source tile plus a 2n halo is staged in LDS once
// (coalesced row-major global reads), the two rectangle passes run separably (rows then columns)
// entirely in LDS, and the tile is written back with coalesced row stores — both operators for the
// price of one read and one write of the mask (2 B/px, HBM-bound).
#include "prl_internal.h"

namespace prl_hip {

namespace {

constexpr int TW = 64;
constexpr int TH = 32;
constexpr int kMaxN = 8;  // halo 2n <= 16 ; larger n falls back to iterated launches of n<=8

template <bool TAKE_MAX>
__device__ __forceinline__ unsigned char mm(unsigned char a, unsigned char b)
{
    return TAKE_MAX ? (a > b ? a : b) : (a < b ? a : b);
}

// dst(rows x cols) = horizontal rect of radius n over src; src has cols + 2n columns.
template <bool TAKE_MAX>
__device__ __forceinline__ void row_pass(const unsigned char* src, int src_pitch, unsigned char* dst,
                                         int dst_pitch, int rows, int cols, int n)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const unsigned char* s = src + r * src_pitch + c;
        unsigned char v = s[0];
        for (int j = 1; j <= 2 * n; ++j) v = mm<TAKE_MAX>(v, s[j]);
        dst[r * dst_pitch + c] = v;
    }
}

// dst(rows x cols) = vertical rect of radius n over src; src has rows + 2n rows.
template <bool TAKE_MAX>
__device__ __forceinline__ void col_pass(const unsigned char* src, int src_pitch, unsigned char* dst,
                                         int dst_pitch, int rows, int cols, int n)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const unsigned char* s = src + r * src_pitch + c;
        unsigned char v = s[0];
        for (int j = 1; j <= 2 * n; ++j) v = mm<TAKE_MAX>(v, s[j * src_pitch]);
        dst[r * dst_pitch + c] = v;
    }
}

// FIRST_MAX = true : closing (dilate, erode) ; false : opening (erode, dilate)
template <bool FIRST_MAX>
__global__ void __launch_bounds__(256) k_morph(PageSet src, PageSetOut dst, int width, int height, int n)
{
    constexpr int PITCH = TW + 4 * kMaxN + 4;  // bytes per LDS row
    __shared__ unsigned char bufA[(TH + 4 * kMaxN) * PITCH];
    __shared__ unsigned char bufB[(TH + 4 * kMaxN) * PITCH];

    const int page = blockIdx.z;
    const uint8_t* in = src.page(page);
    uint8_t* out = dst.page(page);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int h2 = 2 * n;

    // stage source tile with halo 2n; out-of-image pixels take the first operator's neutral value
    const unsigned char neutral1 = FIRST_MAX ? 0 : 255;
    const int sw = TW + 2 * h2, sh = TH + 2 * h2;
    for (int i = threadIdx.x; i < sw * sh; i += blockDim.x) {
        const int r = i / sw, c = i - r * sw;
        const int gy = y0 - h2 + r, gx = x0 - h2 + c;
        unsigned char v = neutral1;
        if (gy >= 0 && gy < height && gx >= 0 && gx < width) v = in[(size_t)gy * src.step + gx];
        bufA[r * PITCH + c] = v;
    }
    __syncthreads();

    // first operator on the tile + halo n
    const int mw = TW + 2 * n, mh = TH + 2 * n;
    row_pass<FIRST_MAX>(bufA, PITCH, bufB, PITCH, sh, mw, n);  // sh rows, mw cols
    __syncthreads();
    col_pass<FIRST_MAX>(bufB, PITCH, bufA, PITCH, mh, mw, n);  // mh rows, mw cols
    __syncthreads();

    // positions outside the image do not exist for the second operator: make them neutral for it
    const unsigned char neutral2 = FIRST_MAX ? 255 : 0;
    for (int i = threadIdx.x; i < mw * mh; i += blockDim.x) {
        const int r = i / mw, c = i - r * mw;
        const int gy = y0 - n + r, gx = x0 - n + c;
        if (gy < 0 || gy >= height || gx < 0 || gx >= width) bufA[r * PITCH + c] = neutral2;
    }
    __syncthreads();

    row_pass<!FIRST_MAX>(bufA, PITCH, bufB, PITCH, mh, TW, n);
    __syncthreads();
    col_pass<!FIRST_MAX>(bufB, PITCH, bufA, PITCH, TH, TW, n);
    __syncthreads();

    for (int i = threadIdx.x; i < TW * TH; i += blockDim.x) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < height && gx < width) out[(size_t)gy * dst.step + gx] = bufA[r * PITCH + c];
    }
}

}  // namespace

// The rectangle of n iterations equals n applications of the 3x3 one, and a closing/opening with
// radius n cannot be split into smaller closings — so radii above kMaxN are rejected here and
// handled by the caller (PRL_ERR_BAD_ARG); the reference's defaults are n in {0, 2}.
int morph_run(int iterations, const PageSet& src, int n_pages, int width, int height,
              const PageSetOut& dst, hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    if (n == 0 || n > kMaxN) {
        set_error_detail("morph_iterations out of range (1.." + std::to_string(kMaxN) + ")");
        return PRL_ERR_BAD_ARG;
    }
    const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, n_pages);
    if (iterations > 0)
        hipLaunchKernelGGL(k_morph<true>, grid, dim3(256), 0, stream, src, dst, width, height, n);
    else
        hipLaunchKernelGGL(k_morph<false>, grid, dim3(256), 0, stream, src, dst, width, height, n);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

}  // namespace prl_hip
