/*
 * prl_oracle_nlm.c — CPU restatement of prl::denoise (src/denoise/denoiseNLM.cpp:29-32), i.e. of
 * cv::fastNlMeansDenoisingColored(src, dst, h = strength, hColor = 3, template 7, search 21).
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  PARITY UNPINNED: the one reference line delegates
 * everything to OpenCV's photo module, which is not in this image; all semantics below are
 * [upstream], restated from SURVEY.md Appendix C:
 *   - FastNlMeansDenoisingInvoker<uchar-vector, int, unsigned, DistSquared, int>: integer SSD over the
 *     7x7 template for each of the 21x21 offsets, `>> 6` binning, weight LUT built from exp() in
 *     float64 with h*h*channels evaluated in float32, fixed-point multiplier 19096, weights below
 *     0.001*19096 zeroed, rounding division by the weight sum.  Integer arithmetic: exact.
 *   - borders: copyMakeBorder(BORDER_REFLECT_101) by 13.
 *   - colour wrapper: LBGR -> Lab (8-bit fixed point RGB2Lab_b), NLM on L with h, NLM on ab with
 *     hColor = 3, Lab -> LBGR (float path of the 3.x series).  The Lab round trip is the
 *     version-dependent part (stated tolerance vs a real OpenCV build: 1 LSB per channel).
 * The oracle computes the SSDs with per-offset running sums (horizontal then vertical 7-tap box of
 * the squared-difference image); the result is the same integer the invoker's incremental update
 * maintains.
 */
#include "prl_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NLM_T 7    /* template window */
#define NLM_S 21   /* search window   */
#define NLM_TH 3
#define NLM_SH 10
#define NLM_BORDER (NLM_TH + NLM_SH)

static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

static int cv_round(double v) { return (int)nearbyint(v); }

/* almost_dist2weight_ of FastNlMeansDenoisingInvoker (DistSquared, WT = int). */
int prl_oracle_nlm_weights(int channels, float h, int32_t* lut, int cap)
{
    /* fixed_point_mult_ = min(INT_MAX / (search*search*255), INT_MAX) */
    const int max_estimate_sum_value = NLM_S * NLM_S * 255;
    const int fixed_point_mult = INT_MAX / max_estimate_sum_value; /* 19096 */
    const int tw_sq = NLM_T * NLM_T;
    int bin_shift = 0;
    while ((1 << bin_shift) < tw_sq) ++bin_shift; /* getNearestPowerOf2(49) = 6 */
    const double mult = ((double)(1 << bin_shift)) / tw_sq;
    const int max_dist = 255 * 255 * channels;
    const int almost_max_dist = (int)(max_dist / mult + 1);
    const float hh = h * h * channels; /* float arithmetic, as in calcWeight */
    const double WEIGHT_THRESHOLD = 0.001;
    int n = almost_max_dist < cap ? almost_max_dist : cap;
    for (int i = 0; i < n; ++i) {
        const double dist = i * mult;
        double w = exp(-dist / hh);
        if (w != w) w = 1.0; /* h == 0 */
        int weight = cv_round(fixed_point_mult * w);
        if (weight < WEIGHT_THRESHOLD * fixed_point_mult) weight = 0;
        lut[i] = weight;
    }
    return n;
}

int prl_oracle_nlm_planes(int channels, float h, const uint8_t* src, size_t src_step, int width,
                          int height, uint8_t* dst, size_t dst_step, int threads)
{
    if (!src || !dst) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels < 1 || channels > 3) return PRL_ERR_BAD_CHANNELS;
    const int ch = channels;
    const int EW = width + 2 * NLM_BORDER, EH = height + 2 * NLM_BORDER;
    uint8_t* E = (uint8_t*)malloc((size_t)EW * EH * ch);
    const int lut_cap = (int)(255.0 * 255.0 * ch / (64.0 / 49.0) + 2);
    int32_t* lut = (int32_t*)calloc((size_t)lut_cap, sizeof(int32_t));
    if (!E || !lut) {
        free(E);
        free(lut);
        return PRL_ERR_NOMEM;
    }
    const int lut_n = prl_oracle_nlm_weights(ch, h, lut, lut_cap);
    for (int y = 0; y < EH; ++y) {
        const uint8_t* s = src + (size_t)reflect101(y - NLM_BORDER, height) * src_step;
        uint8_t* d = E + (size_t)y * EW * ch;
        for (int x = 0; x < EW; ++x) {
            const int sx = reflect101(x - NLM_BORDER, width);
            for (int c = 0; c < ch; ++c) d[x * ch + c] = s[sx * ch + c];
        }
    }
#define EPIX(y, x) (E + ((size_t)((y) + NLM_BORDER) * EW + ((x) + NLM_BORDER)) * ch)

    if (threads < 1) threads = 1;
    const int band = 32;
    const int n_bands = (height + band - 1) / band;
    int status = PRL_OK;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int b = 0; b < n_bands; ++b) {
        const int r0 = b * band, r1 = (r0 + band < height) ? r0 + band : height;
        const int rows = r1 - r0;
        int32_t* est = (int32_t*)calloc((size_t)rows * width * ch, sizeof(int32_t));
        int32_t* wsum = (int32_t*)calloc((size_t)rows * width, sizeof(int32_t));
        int32_t* hs = (int32_t*)malloc((size_t)(rows + 2 * NLM_TH) * width * sizeof(int32_t));
        int32_t* d2 = (int32_t*)malloc((size_t)(width + 2 * NLM_TH) * sizeof(int32_t));
        if (!est || !wsum || !hs || !d2) {
            status = PRL_ERR_NOMEM;
            free(est);
            free(wsum);
            free(hs);
            free(d2);
            continue;
        }
        for (int dy = -NLM_SH; dy <= NLM_SH; ++dy)
            for (int dx = -NLM_SH; dx <= NLM_SH; ++dx) {
                /* hs(r, j) = sum_{tx=-3..3} sum_c (E(y, j+tx) - E(y+dy, j+tx+dx))^2 for y = r0-3+r */
                for (int r = 0; r < rows + 2 * NLM_TH; ++r) {
                    const int y = r0 - NLM_TH + r;
                    for (int x = -NLM_TH; x < width + NLM_TH; ++x) {
                        const uint8_t* a = EPIX(y, x);
                        const uint8_t* bb = EPIX(y + dy, x + dx);
                        int s = 0;
                        for (int c = 0; c < ch; ++c) {
                            const int df = (int)a[c] - (int)bb[c];
                            s += df * df;
                        }
                        d2[x + NLM_TH] = s;
                    }
                    int32_t* hrow = hs + (size_t)r * width;
                    int run = 0;
                    for (int k = 0; k < NLM_T; ++k) run += d2[k];
                    hrow[0] = run;
                    for (int j = 1; j < width; ++j) {
                        run += d2[j + NLM_T - 1] - d2[j - 1];
                        hrow[j] = run;
                    }
                }
                for (int i = 0; i < rows; ++i) {
                    const uint8_t* q = EPIX(r0 + i + dy, dx);
                    int32_t* er = est + (size_t)i * width * ch;
                    int32_t* wr = wsum + (size_t)i * width;
                    for (int j = 0; j < width; ++j) {
                        int D = 0;
                        for (int k = 0; k < NLM_T; ++k) D += hs[(size_t)(i + k) * width + j];
                        const int idx = D >> 6; /* almost_template_window_size_sq_bin_shift_ */
                        const int wgt = idx < lut_n ? lut[idx] : 0;
                        wr[j] += wgt;
                        for (int c = 0; c < ch; ++c) er[j * ch + c] += wgt * (int)q[j * ch + c];
                    }
                }
            }
        for (int i = 0; i < rows; ++i) {
            uint8_t* o = dst + (size_t)(r0 + i) * dst_step;
            for (int j = 0; j < width; ++j) {
                const unsigned ws = (unsigned)wsum[(size_t)i * width + j];
                for (int c = 0; c < ch; ++c) {
                    /* divByWeightsSum: (unsigned(est) + wsum/2) / wsum, then saturate_cast<uchar> */
                    const unsigned e = (unsigned)est[((size_t)i * width + j) * ch + c];
                    const unsigned v = (e + ws / 2) / ws;
                    o[j * ch + c] = (uint8_t)(v > 255 ? 255 : v);
                }
            }
        }
        free(est);
        free(wsum);
        free(hs);
        free(d2);
    }
#undef EPIX
    free(E);
    free(lut);
    return status;
}

/* ---------------------------------------------------------------------------------------------
 * 8-bit LBGR <-> Lab, OpenCV 3.x color.cpp [upstream, restated from memory of that source]:
 * RGB2Lab_b with linearGammaTab_b (no gamma: "L" prefix), LabCbrtTab_b, lab_shift = 12,
 * gamma_shift = 3, lab_shift2 = 15; Lab2RGB_b through the float Lab2RGB_f with gamma disabled.
 * cvCbrt is replaced by cbrtf: table entries may differ from a real OpenCV build in the last unit
 * for a handful of indices (part of the stated 1 LSB colour round-trip tolerance).
 * ------------------------------------------------------------------------------------------- */
#define LAB_SHIFT 12
#define GAMMA_SHIFT 3
#define LAB_SHIFT2 (LAB_SHIFT + GAMMA_SHIFT)
#define LAB_CBRT_TAB_SIZE_B (256 * 3 / 2 * (1 << GAMMA_SHIFT))
#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static const float k_sRGB2XYZ_D65[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                                        0.072169f, 0.019334f, 0.119193f, 0.950227f};
static const float k_XYZ2sRGB_D65[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f,
                                        0.041556f, 0.055648f, -0.204043f, 1.057311f};
static const float k_D65[3] = {0.950456f, 1.f, 1.088754f};

static uint16_t g_cbrt_tab_b[LAB_CBRT_TAB_SIZE_B];
static int g_lab_coeffs[9];
static float g_inv_coeffs[9];
static int g_lab_init = 0;

static uint16_t sat_u16(float v)
{
    const int iv = cv_round((double)v);
    return (uint16_t)(iv < 0 ? 0 : (iv > 65535 ? 65535 : iv));
}

void prl_oracle_lab_tables(const uint16_t** cbrt_tab, const int** fwd_coeffs, const float** inv_coeffs);

static void lab_init(void)
{
#ifdef _OPENMP
#pragma omp critical(prl_lab_init)
#endif
    if (!g_lab_init) {
        for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; ++i) {
            const float x = i * (1.f / (255.f * (1 << GAMMA_SHIFT)));
            g_cbrt_tab_b[i] =
                sat_u16((1 << LAB_SHIFT2) * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cbrtf(x)));
        }
        /* blueIdx = 0 (BGR): coefficient columns swapped so that src[0] = B meets the blue column */
        const float scale[3] = {(1 << LAB_SHIFT) / k_D65[0], (float)(1 << LAB_SHIFT), (1 << LAB_SHIFT) / k_D65[2]};
        for (int i = 0; i < 3; ++i) {
            g_lab_coeffs[i * 3 + 2] = cv_round(k_sRGB2XYZ_D65[i * 3] * scale[i]);
            g_lab_coeffs[i * 3 + 1] = cv_round(k_sRGB2XYZ_D65[i * 3 + 1] * scale[i]);
            g_lab_coeffs[i * 3 + 0] = cv_round(k_sRGB2XYZ_D65[i * 3 + 2] * scale[i]);
        }
        for (int i = 0; i < 3; ++i) {
            g_inv_coeffs[i + 2 * 3] = k_XYZ2sRGB_D65[i] * k_D65[i];     /* (blueIdx^2)*3 : R row last */
            g_inv_coeffs[i + 3] = k_XYZ2sRGB_D65[i + 3] * k_D65[i];
            g_inv_coeffs[i + 0 * 3] = k_XYZ2sRGB_D65[i + 6] * k_D65[i]; /* blueIdx*3 : B row first */
        }
        g_lab_init = 1;
    }
}

void prl_oracle_lab_tables(const uint16_t** cbrt_tab, const int** fwd_coeffs, const float** inv_coeffs)
{
    lab_init();
    if (cbrt_tab) *cbrt_tab = g_cbrt_tab_b;
    if (fwd_coeffs) *fwd_coeffs = g_lab_coeffs;
    if (inv_coeffs) *inv_coeffs = g_inv_coeffs;
}

static uint8_t sat_u8_int(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void prl_oracle_lbgr2lab(const uint8_t* bgr, size_t src_step, int width, int height, int channels,
                         uint8_t* lab, size_t dst_step)
{
    lab_init();
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const int* C = g_lab_coeffs;
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = bgr + (size_t)y * src_step;
        uint8_t* d = lab + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x, s += channels, d += 3) {
            const int R = s[0] << GAMMA_SHIFT, G = s[1] << GAMMA_SHIFT, B = s[2] << GAMMA_SHIFT; /* linearGammaTab_b */
            const int fX = g_cbrt_tab_b[CV_DESCALE(R * C[0] + G * C[1] + B * C[2], LAB_SHIFT)];
            const int fY = g_cbrt_tab_b[CV_DESCALE(R * C[3] + G * C[4] + B * C[5], LAB_SHIFT)];
            const int fZ = g_cbrt_tab_b[CV_DESCALE(R * C[6] + G * C[7] + B * C[8], LAB_SHIFT)];
            const int L = CV_DESCALE(Lscale * fY + Lshift, LAB_SHIFT2);
            const int a = CV_DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
            const int b = CV_DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
            d[0] = sat_u8_int(L);
            d[1] = sat_u8_int(a);
            d[2] = sat_u8_int(b);
        }
    }
}

static inline float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

void prl_oracle_lab2lbgr(const uint8_t* lab, size_t src_step, int width, int height, uint8_t* bgr,
                         size_t dst_step, int channels)
{
    lab_init();
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const float* C = g_inv_coeffs;
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = lab + (size_t)y * src_step;
        uint8_t* d = bgr + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x, s += 3, d += channels) {
            /* Lab2RGB_b: L*100/255, a-128, b-128, then Lab2RGB_f (srgb = false) */
            const float li = s[0] * (100.f / 255.f);
            const float ai = (float)(s[1] - 128);
            const float bi = (float)(s[2] - 128);
            float Y, fy;
            if (li <= lThresh) {
                Y = li / 903.3f;
                fy = 7.787f * Y + 16.0f / 116.0f;
            } else {
                fy = (li + 16.0f) / 116.0f;
                Y = fy * fy * fy;
            }
            float fxz[2] = {ai / 500.0f + fy, fy - bi / 200.0f};
            for (int j = 0; j < 2; ++j) {
                if (fxz[j] <= fThresh) fxz[j] = (fxz[j] - 16.0f / 116.0f) / 7.787f;
                else fxz[j] = fxz[j] * fxz[j] * fxz[j];
            }
            const float X = fxz[0], Z = fxz[1];
            float c0 = C[0] * X + C[1] * Y + C[2] * Z;
            float c1 = C[3] * X + C[4] * Y + C[5] * Z;
            float c2 = C[6] * X + C[7] * Y + C[8] * Z;
            c0 = clip01(c0);
            c1 = clip01(c1);
            c2 = clip01(c2);
            d[0] = sat_u8_int(cv_round((double)(c0 * 255.f)));
            d[1] = sat_u8_int(cv_round((double)(c1 * 255.f)));
            d[2] = sat_u8_int(cv_round((double)(c2 * 255.f)));
            if (channels == 4) d[3] = 255;
        }
    }
}

/* cv::fastNlMeansDenoisingColored(src, dst, h, hColor = 3, 7, 21) — denoiseNLM.cpp:31 */
int prl_oracle_denoise(int channels, float strength, const uint8_t* src, size_t src_step, int width,
                       int height, uint8_t* dst, size_t dst_step, int threads)
{
    if (!src || !dst) return PRL_ERR_BAD_ARG;
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS; /* "Type of input image should be CV_8UC3 or CV_8UC4!" */
    const size_t n = (size_t)width * height;
    uint8_t* lab = (uint8_t*)malloc(n * 3);
    uint8_t* l = (uint8_t*)malloc(n);
    uint8_t* ab = (uint8_t*)malloc(n * 2);
    uint8_t* l2 = (uint8_t*)malloc(n);
    uint8_t* ab2 = (uint8_t*)malloc(n * 2);
    if (!lab || !l || !ab || !l2 || !ab2) {
        free(lab); free(l); free(ab); free(l2); free(ab2);
        return PRL_ERR_NOMEM;
    }
    prl_oracle_lbgr2lab(src, src_step, width, height, channels, lab, (size_t)width * 3);
    for (size_t i = 0; i < n; ++i) { /* mixChannels: 0->L, 1,2->ab */
        l[i] = lab[3 * i];
        ab[2 * i] = lab[3 * i + 1];
        ab[2 * i + 1] = lab[3 * i + 2];
    }
    int st = prl_oracle_nlm_planes(1, strength, l, (size_t)width, width, height, l2, (size_t)width, threads);
    if (st == PRL_OK)
        st = prl_oracle_nlm_planes(2, 3.0f, ab, (size_t)width * 2, width, height, ab2, (size_t)width * 2, threads);
    if (st == PRL_OK) {
        for (size_t i = 0; i < n; ++i) {
            lab[3 * i] = l2[i];
            lab[3 * i + 1] = ab2[2 * i];
            lab[3 * i + 2] = ab2[2 * i + 1];
        }
        prl_oracle_lab2lbgr(lab, (size_t)width * 3, width, height, dst, dst_step, channels);
    }
    free(lab); free(l); free(ab); free(l2); free(ab2);
    return st;
}
