/*
 * prl_oracle_lv.c — CPU restatement of prl::binarizeByLocalVariances and prl::binarizeByLocalVariancesWithoutFilters
 * (SURVEY.md §8f rank 4b).
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  PARITY STATUS: **parity unpinned**, and for the first function
 * additionally only defined up to the float32 transcendental functions: the reference calls cv::log and cv::exp, which
 * are OpenCV's own table-driven float32 routines (not libm), so two OpenCV builds agree with each other but nothing
 * outside OpenCV reproduces their last bit.  This restatement uses logf / expf; tests compare the device result with
 * it under a stated tolerance (mismatching pixels <= 1e-3 of the page), not bit for bit.
 *
 * Reference: src/binarizations/binarizeByLocalVariances.cpp:13-145 (with filters), :148-292 (without),
 *            src/imageLibCommon.cpp:397-466 (MatToLocalVarianceMap, kernelSize 3).
 * Both need a 3-channel 8-bit image: :37-38 splits the variance map into three planes and reads all three
 * (a 1-channel input indexes past the end of the vector - undefined in the reference, an error here).
 *
 * OpenCV arithmetic restated [upstream]:
 *   MatToLocalVarianceMap   float32 box sums of p and p*p over 3x3, BORDER_REPLICATE (every value an integer below 2^24,
 *                           so the sums are exact whatever the accumulation order); (9 * sum(p^2) - sum(p)^2) exact;
 *                           one float32 multiplication by (float)(1 / (9.f * 9.f)); values below 0.01f become 0.01f
 *                           (imageLibCommon.cpp:447-465)
 *   cv::filter2D, float32   the contrast kernel {0,-1,0; -1,16,-1; 0,-1,0} (:60-80), BORDER_REPLICATE: direct path, the
 *                           five non-zero taps in row-major order, s = 0; s += k[i] * x[i], one rounding per multiply and
 *                           per add (no FMA) - ASSUMED accumulation order (OpenCV's FilterVec_32f; FMA builds differ in
 *                           the last bit)
 *   Vec3f arithmetic        (max - min) in float32; / 2 and * coeff through double with one rounding to float32 each (:83-85)
 *   cv::log / cv::exp       logf / expf here (see above)
 *   convertTo(CV_32F, a, b) float32 x * (float)a + (float)b, multiply then add (:116-119)
 *   cv::pow(x, 2.0)         x * x; other exponents powf (:122)
 *   cv::convertScaleAbs     saturate_cast<uchar>(|x * (float)scale|), cvRound = round-half-even (:124, :136)
 *   cv::adaptiveThreshold   15 x 15 box mean of the 8-bit map, BORDER_REPLICATE, mean = cvRound(sum * (1.0 / 225)),
 *                           result 127 where src - mean > 0 (:126-127)
 *   cv::mean                float64 sum / count (:131)
 *   Mat - Mat on 8-bit      saturating; 255 * (a > b) saturates to 255 (:139, :42-47)
 * The second function is scalar Vec3f code in the reference itself (no OpenCV arithmetic beyond Vec3f operators):
 * integer-valued float32 sums, variance as above, contrast sum ((((((((0*a + -1*b) + 0*c) + -1*d) + 16*e) + -1*f) + 0*g)
 * + -1*h) + 0*i) in float32 over the BORDER_REFLECT-padded map (= replicate for a 1-pixel border) (:244-286).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "prl_oracle.h"

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static uint8_t sat_u8_f(float v)
{
    const float r = rintf(v); /* cvRound */
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

/* MatToLocalVarianceMap(image, map, 3) on an 8UC3 image: var is h x w x 3 float32. */
void prl_oracle_local_variance_map(const uint8_t* bgr, size_t step, int width, int height, float* var)
{
    const float area = 9.0f, vmin = 0.01f;
    const float scale = (float)(1 / (area * area));
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) {
                int s = 0, q = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int p = bgr[(size_t)clampi(y + dy, 0, height - 1) * step + (size_t)clampi(x + dx, 0, width - 1) * 3 + c];
                        s += p;
                        q += p * p;
                    }
                float v = (float)(9 * q - s * s) * scale;
                if (!(v > vmin)) v = vmin; /* THRESH_TOZERO + THRESH_BINARY_INV with the same threshold */
                var[((size_t)y * width + x) * 3 + c] = v;
            }
}

static void min_max3(const float* var, size_t n, float mn[3], float mx[3])
{
    for (int c = 0; c < 3; ++c) { mn[c] = var[c]; mx[c] = var[c]; }
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const float v = var[i * 3 + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
}

/* globalMinMaxHalfDist * coeff (binarizeByLocalVariances.cpp:83-85, :240-242) */
static void variance_thresholds(const float mn[3], const float mx[3], double coeff, float thr[3])
{
    for (int c = 0; c < 3; ++c) {
        const float dist = mx[c] - mn[c];
        const float half = (float)((double)dist * (1. / 2));
        thr[c] = (float)((double)half * coeff);
    }
}

int prl_oracle_binarize_lv(const uint8_t* bgr, size_t step, int width, int height, double coeff, int min_result_variance,
                           double gamma, uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0 || !bgr) return PRL_ERR_EMPTY; /* :16-19 */
    if (!dst || step < (size_t)width * 3 || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    const size_t n = (size_t)width * height;
    float* var = (float*)malloc(n * 3 * sizeof(float));
    float* L = (float*)malloc(n * sizeof(float));
    uint8_t* G = (uint8_t*)malloc(n);
    prl_oracle_local_variance_map(bgr, step, width, height, var);
    float mn[3], mx[3], thr[3];
    min_max3(var, n, mn, mx);
    variance_thresholds(mn, mx, coeff, thr);
    /* log map, its range and mean (:104-131) */
    float lmin = 0, lmax = 0;
    double lsum = 0;
    for (size_t i = 0; i < n; ++i) {
        const float l = (logf(var[i * 3]) + logf(var[i * 3 + 1])) + logf(var[i * 3 + 2]);
        L[i] = l;
        if (i == 0 || l < lmin) lmin = l;
        if (i == 0 || l > lmax) lmax = l;
        lsum += l;
    }
    const double range = (double)lmax - (double)lmin;
    const float ga = (float)(1.0 / range), gb = (float)(-(double)lmin / range);
    const float lmean = (float)(lsum / (double)n);
    for (size_t i = 0; i < n; ++i) {
        const float t = L[i] * ga + gb;
        const float tg = gamma == 2.0 ? t * t : powf(t, (float)gamma);
        G[i] = sat_u8_f(fabsf(tg * 255.0f));
    }
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const size_t i = (size_t)y * width + x;
            /* result1 (:42-47): any channel above 10 */
            const int r1 = var[i * 3] > 10.0f || var[i * 3 + 1] > 10.0f || var[i * 3 + 2] > 10.0f;
            /* result2 (:80-97): contrast-filtered variance above the per-channel threshold */
            int r2 = 0;
            for (int c = 0; c < 3; ++c) {
                const float up = var[((size_t)clampi(y - 1, 0, height - 1) * width + x) * 3 + c];
                const float lf = var[((size_t)y * width + clampi(x - 1, 0, width - 1)) * 3 + c];
                const float ce = var[i * 3 + c];
                const float rt = var[((size_t)y * width + clampi(x + 1, 0, width - 1)) * 3 + c];
                const float dn = var[((size_t)clampi(y + 1, 0, height - 1) * width + x) * 3 + c];
                float s = 0.0f;
                s += -1.0f * up;
                s += -1.0f * lf;
                s += 16.0f * ce;
                s += -1.0f * rt;
                s += -1.0f * dn;
                r2 |= s > thr[c];
            }
            /* adaptiveThreshold(G, 127, MEAN_C, BINARY, 15, 0) (:126-127) */
            int bs = 0;
            for (int dy = -7; dy <= 7; ++dy)
                for (int dx = -7; dx <= 7; ++dx) bs += G[(size_t)clampi(y + dy, 0, height - 1) * width + clampi(x + dx, 0, width - 1)];
            const int mean = (int)lrint((double)bs * (1. / 225)); /* saturate_cast<uchar>: always within 0..255 */
            const int a = ((int)G[i] - mean > 0) ? 127 : 0;
            /* noise term (:130-136) */
            const float d = L[i] - lmean;
            const float e = expf(-(d * d) * 0.5f);
            const int nz = sat_u8_f(fabsf(e * 127.0f));
            int diff = a - nz;
            if (diff < 0) diff = 0;
            const int r3 = diff > min_result_variance;
            dst[(size_t)y * dst_step + x] = (r1 && r2 && r3) ? 255 : 0;
        }
    free(var);
    free(L);
    free(G);
    return PRL_OK;
}

int prl_oracle_binarize_lv_nofilters(const uint8_t* bgr, size_t step, int width, int height, double coeff,
                                     int min_result_variance, uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0 || !bgr) return PRL_ERR_EMPTY; /* :151-154 */
    if (!dst || step < (size_t)width * 3 || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    const size_t n = (size_t)width * height;
    float* var = (float*)malloc(n * 3 * sizeof(float));
    prl_oracle_local_variance_map(bgr, step, width, height, var); /* same arithmetic as the scalar loop :176-236 */
    float mn[3], mx[3], thr[3];
    min_max3(var, n, mn, mx); /* (the initial -1 / 1000000 of :172-173 never win: 0.01 <= variance <= 16256.25) */
    variance_thresholds(mn, mx, coeff, thr);
    const float mask[9] = {0, -1, 0, -1, 16, -1, 0, -1, 0};
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const size_t i = (size_t)y * width + x;
            float mv = var[i * 3] > var[i * 3 + 1] ? var[i * 3] : var[i * 3 + 1];
            mv = mv > var[i * 3 + 2] ? mv : var[i * 3 + 2];
            const int r1 = mv > (float)min_result_variance;
            int r2 = 0;
            for (int c = 0; c < 3; ++c) {
                float s = 0.0f;
                int first = 1;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float v = var[((size_t)clampi(y + dy, 0, height - 1) * width + clampi(x + dx, 0, width - 1)) * 3 + c];
                        const float t = v * mask[(dy + 1) * 3 + dx + 1];
                        s = first ? t : s + t; /* neighbor0 + neighbor1 + ... left to right */
                        first = 0;
                    }
                r2 |= s > thr[c];
            }
            dst[(size_t)y * dst_step + x] = (r1 && r2) ? 255 : 0;
        }
    free(var);
    return PRL_OK;
}
