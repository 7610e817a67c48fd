/*
 * prl_oracle_thin.c — CPU restatement of prl::thinZhangSuen (src/thinning/thinZhangSuen.cpp:15-108) and
 * prl::thinGuoHall (src/thinning/thinGuoHall.cpp:15-107).
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  These two functions contain no OpenCV arithmetic — cv::Mat is
 * only a container there — so the restatement is the reference's own integer logic, line by line.  The
 * reference cannot be compiled here (it includes OpenCV headers), hence still "parity unpinned" formally, but
 * nothing [upstream] is involved apart from `&= 1`, absdiff/countNonZero (any change?) and `* 255`.
 */
#include "prl_oracle.h"

#include <stdlib.h>
#include <string.h>

/* thinZhangSuenIteration (thinZhangSuen.cpp:15-55) / thinGuoHallIteration (thinGuoHall.cpp:15-54) on a
 * continuous rows x cols image with values 0/1; iteration = 0 or 1. */
static void thin_iteration(int method, uint8_t* im, int rows, int cols, int iteration, uint8_t* marker)
{
    memset(marker, 0, (size_t)rows * cols);
    for (int i = 1; i < rows - 1; ++i) {
        for (int j = 1; j < cols - 1; ++j) {
            const uint8_t p2 = im[(size_t)(i - 1) * cols + (j - 0)];
            const uint8_t p3 = im[(size_t)(i - 1) * cols + (j + 1)];
            const uint8_t p4 = im[(size_t)(i - 0) * cols + (j + 1)];
            const uint8_t p5 = im[(size_t)(i + 1) * cols + (j + 1)];
            const uint8_t p6 = im[(size_t)(i + 1) * cols + (j - 0)];
            const uint8_t p7 = im[(size_t)(i + 1) * cols + (j - 1)];
            const uint8_t p8 = im[(size_t)(i - 0) * cols + (j - 1)];
            const uint8_t p9 = im[(size_t)(i - 1) * cols + (j - 1)];
            if (method == 0) {
                /* thinZhangSuen.cpp:37-51 */
                const int A = (p2 == 0 && p3 == 1) + (p3 == 0 && p4 == 1) + (p4 == 0 && p5 == 1) + (p5 == 0 && p6 == 1) +
                              (p6 == 0 && p7 == 1) + (p7 == 0 && p8 == 1) + (p8 == 0 && p9 == 1) + (p9 == 0 && p2 == 1);
                const int B = p2 + p3 + p4 + p5 + p6 + p7 + p8 + p9;
                const int m1 = iteration == 0 ? (p2 * p4 * p6) : (p2 * p4 * p8);
                const int m2 = iteration == 0 ? (p4 * p6 * p8) : (p2 * p6 * p8);
                if (A == 1 && (B >= 2 && B <= 6) && m1 == 0 && m2 == 0) marker[(size_t)i * cols + j] = 1;
            } else {
                /* thinGuoHall.cpp:40-50 */
                const int C = ((!p2) & (p3 | p4)) + ((!p4) & (p5 | p6)) + ((!p6) & (p7 | p8)) + ((!p8) & (p9 | p2));
                const int N1 = (p9 | p2) + (p3 | p4) + (p5 | p6) + (p7 | p8);
                const int N2 = (p2 | p3) + (p4 | p5) + (p6 | p7) + (p8 | p9);
                const int N = N1 < N2 ? N1 : N2;
                const int m = iteration == 0 ? ((p6 | p7 | !p9) & p8) : ((p2 | p3 | !p5) & p4);
                if (C == 1 && (N >= 2 && N <= 3) && (m == 0)) marker[(size_t)i * cols + j] = 1;
            }
        }
    }
    /* imageUnderProcessing &= ~marker   (thinZhangSuen.cpp:54) */
    for (size_t k = 0; k < (size_t)rows * cols; ++k) im[k] &= (uint8_t)~marker[k];
}

/* The body of prl::thinZhangSuen / prl::thinGuoHall after cvtColor (thinZhangSuen.cpp:83-107):
 * `&= 1`, iterate both sub-iterations until a whole pass changes nothing, `* 255`.  method: 0 Zhang-Suen, 1 Guo-Hall. */
int prl_oracle_thin(int method, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                    size_t dst_step, int* passes_out)
{
    if (!src || !dst) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (method != 0 && method != 1) return PRL_ERR_BAD_ARG;
    const size_t n = (size_t)width * height;
    uint8_t* im = (uint8_t*)malloc(n);
    uint8_t* prev = (uint8_t*)calloc(n, 1);
    uint8_t* marker = (uint8_t*)malloc(n);
    if (!im || !prev || !marker) {
        free(im); free(prev); free(marker);
        return PRL_ERR_NOMEM;
    }
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) im[(size_t)y * width + x] = src[(size_t)y * src_step + x] & 1; /* :83 */
    int passes = 0, changed;
    do {
        thin_iteration(method, im, height, width, 0, marker);
        thin_iteration(method, im, height, width, 1, marker);
        changed = memcmp(im, prev, n) != 0; /* absdiff + countNonZero(diff) > 0  (:93-96) */
        memcpy(prev, im, n);
        ++passes;
    } while (changed);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) dst[(size_t)y * dst_step + x] = (uint8_t)(im[(size_t)y * width + x] * 255); /* :100-106 */
    if (passes_out) *passes_out = passes;
    free(im); free(prev); free(marker);
    return PRL_OK;
}
