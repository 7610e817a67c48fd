/*
 * prl_oracle.h — CPU restatement of PRLib's local-adaptive binarizers and NL-means stage.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under prlib_amd/ may include, link or call this.  It is used
 * by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (zamazan4ik/PRLib) cannot be built here: every
 * hot-path file needs OpenCV (src/binarizations/binarizeSauvola.cpp:29, src/denoise/denoiseNLM.cpp:27)
 * and the library links Leptonica (CMakeLists.txt:17,33); neither exists in this image, and the
 * reference ships no tests or golden outputs.  This file therefore restates, operation by
 * operation, what the reference's source asks OpenCV to compute, with the canonical choices of
 * SURVEY.md Appendix A/B (IEEE binary64, one rounding per written operation, no FMA contraction,
 * direct 4-tap filter2D in row-major tap order, cvRound = round-half-even, NaN -> 0 on u8 cast).
 * It is cross-checked against an independently written numpy model (oracle/numpy_model.py) and
 * hand-derived known answers in tests/.
 */
#ifndef PRL_ORACLE_H_
#define PRL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>
#include "../include/prl_hip.h" /* shares prl_method / prl_binarize_params / prl_status */

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry (w, half, padded size, output size) and argument validation, binarizeSauvola.cpp:38-66. */
int prl_oracle_binarize_geometry(const prl_binarize_params* p, int width, int height,
                                 prl_binarize_geometry* out);

/* cv::copyMakeBorder(BORDER_REPLICATE) as called at binarizeSauvola.cpp:65. */
void prl_oracle_pad_replicate(const uint8_t* src, size_t src_step, int width, int height, int half,
                              uint8_t* dst, size_t dst_step);

/* cv::integral(src, sum, sqsum, CV_64F) with the zero row/column dropped (binarizeSauvola.cpp:72-77):
 * ii/iq are padded_h x padded_w doubles, inclusive prefix sums of P and P*P. */
void prl_oracle_integrals(const uint8_t* padded, size_t step, int pw, int ph, double* ii, double* iq);

/* Local mean / deviation planes exactly as the two filter2D calls + mul/-=/sqrt produce them
 * (binarizeSauvola.cpp:83-110).  mean/dev are out_h x out_w doubles (either may be NULL). */
int prl_oracle_mean_dev(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                        int width, int height, double* mean, double* dev);

/* Threshold plane T (float64, before the CV_8U cast) for any of the five methods. */
int prl_oracle_threshold_plane(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                               int width, int height, double* T);

/* saturate_cast<uchar>(double) as convertTo(CV_8U) applies it (binarizeSauvola.cpp:119). */
uint8_t prl_oracle_sat_u8(double v);

/* The whole function body after cvtColor: pad, integrals, thresholds, compare, morphology. */
int prl_oracle_binarize(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                        int width, int height, uint8_t* dst, size_t dst_step);

/* n pages with `threads` OpenMP threads (1 = serial); used as the timed CPU baseline. */
int prl_oracle_binarize_batch(const prl_binarize_params* p, int n_pages,
                              const uint8_t* src, size_t src_page_stride, size_t src_step,
                              int width, int height,
                              uint8_t* dst, size_t dst_page_stride, size_t dst_step, int threads);

/* cv::dilate/cv::erode with the default 3x3 kernel iterated n times = (2n+1)^2 rect, border ignored
 * (binarizeSauvola.cpp:125-134).  n>0 closing (dilate then erode), n<0 opening. */
void prl_oracle_morph(int morph_iterations, const uint8_t* src, size_t src_step, int width, int height,
                      uint8_t* dst, size_t dst_step);

/* cv::cvtColor(BGR2GRAY) 8-bit, 14-bit fixed point (binarizeSauvola.cpp:51). */
void prl_oracle_bgr2gray(const uint8_t* bgr, size_t src_step, int width, int height, int channels,
                         uint8_t* gray, size_t dst_step);

/* Otsu threshold of an 8-bit image (cv::threshold(..., THRESH_BINARY|THRESH_OTSU), deskew.cpp:224):
 * returns the threshold; dst (may be NULL) gets p > thr ? 255 : 0. */
int prl_oracle_otsu(const uint8_t* src, size_t src_step, int width, int height,
                    uint8_t* dst, size_t dst_step);

/* ---- NL-means (reference line: src/denoise/denoiseNLM.cpp:31) ---- */

/* Weight LUT of FastNlMeansDenoisingInvoker for `channels` interleaved planes: returns the number
 * of entries written (<= cap); entries beyond it are zero. */
int prl_oracle_nlm_weights(int channels, float h, int32_t* lut, int cap);

/* fastNlMeansDenoising core on 1/2/3 interleaved u8 planes, template 7, search 21. */
int prl_oracle_nlm_planes(int channels, float h, const uint8_t* src, size_t src_step,
                          int width, int height, uint8_t* dst, size_t dst_step, int threads);

/* 8-bit LBGR<->Lab (cv::cvtColor COLOR_LBGR2Lab / COLOR_Lab2LBGR), canonical integer tables. */
void prl_oracle_lbgr2lab(const uint8_t* bgr, size_t src_step, int width, int height, int channels,
                         uint8_t* lab, size_t dst_step);
void prl_oracle_lab2lbgr(const uint8_t* lab, size_t src_step, int width, int height,
                         uint8_t* bgr, size_t dst_step, int channels);

/* prl::denoise = fastNlMeansDenoisingColored(src, dst, strength) on BGR/BGRA. */
int prl_oracle_denoise(int channels, float strength, const uint8_t* src, size_t src_step,
                       int width, int height, uint8_t* dst, size_t dst_step, int threads);

/* ---- thinning (src/thinning/thinZhangSuen.cpp:57-108, thinGuoHall.cpp:56-107) ---- */
/* method 0 = Zhang-Suen, 1 = Guo-Hall; src is the 1-channel image after cvtColor; passes_out (optional) gets the
 * number of do-while passes the reference loop makes. */
int prl_oracle_thin(int method, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                    size_t dst_step, int* passes_out);

/* ---- backgroundNormalization (src/backgroundNormalization.cpp:36-61; prl_oracle_bgnorm.c) ---- */
/* channels 1 -> 1-channel result; 3 or 4 -> 3-channel result (src/formatConvert.cpp:111-218). */
int  prl_oracle_bgnorm_out_channels(int channels);
void prl_oracle_bgnorm_map_size(int width, int height, int* map_w, int* map_h);
void prl_oracle_bgnorm_fgmask(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* fg);
int  prl_oracle_bgnorm_bgmap(int channels, int c, const uint8_t* src, size_t src_step, int width, int height,
                             const uint8_t* fg, uint8_t* map);
void prl_oracle_bgnorm_blockconv(const uint8_t* map, int w, int h, uint8_t* out);
int  prl_oracle_bgnorm_invmap(const uint8_t* map, int w, int h, uint16_t* inv);
int  prl_oracle_bgnorm(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                       size_t dst_step);

/* ---- deskew / rotate (src/deskew/deskew.cpp:139-251, src/rotate.cpp:35-72; prl_oracle_deskew.c) ---- */
/* cv::HoughLinesP(image, lines, 1, CV_PI/180, threshold, lineLength, lineGap): returns the number of segments found,
 * the first `cap` of them as (x0, y0, x1, y1). */
int    prl_oracle_houghp(const uint8_t* image, size_t step, int width, int height, int threshold, int lineLength,
                         int lineGap, int32_t* lines, int cap);
double prl_oracle_vote_angle(const int32_t* lines, int nb_lines);
double prl_oracle_find_angle(const uint8_t* bin, size_t step, int width, int height, int* n_lines_out);
int    prl_oracle_rotate_kind(double angle);
void   prl_oracle_rotate_size(int width, int height, double angle, int* out_w, int* out_h);
void   prl_oracle_rotate_matrix(int width, int height, double angle, double M[6]);
int    prl_oracle_rotate(int channels, const uint8_t* src, size_t src_step, int width, int height, double angle,
                         uint8_t* dst, size_t dst_step);
int    prl_oracle_deskew(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                         size_t dst_step, int* out_w, int* out_h, double* angle_out, int* thr_out, int* n_lines_out);

/* ---- binarizeByLocalVariances (src/binarizations/binarizeByLocalVariances.cpp; prl_oracle_lv.c) ---- */
void prl_oracle_local_variance_map(const uint8_t* bgr, size_t step, int width, int height, float* var);
int  prl_oracle_binarize_lv(const uint8_t* bgr, size_t step, int width, int height, double coeff, int min_result_variance,
                            double gamma, uint8_t* dst, size_t dst_step);
int  prl_oracle_binarize_lv_nofilters(const uint8_t* bgr, size_t step, int width, int height, double coeff,
                                      int min_result_variance, uint8_t* dst, size_t dst_step);

#ifdef __cplusplus
}
#endif
#endif
