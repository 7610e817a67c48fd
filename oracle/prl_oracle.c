/*
 * prl_oracle.c — CPU restatement of PRLib's five local-adaptive binarizers.
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  PARITY UNPINNED against the real reference: the
 * reference needs OpenCV + Leptonica, absent from this image, and ships no golden outputs.
 *
 * Every function cites the reference lines it restates (paths relative to the PRLib tree).
 * Statements about OpenCV internals are marked [upstream]: they follow SURVEY.md Appendix B.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (the Makefile enforces it).  The float64
 * sequence below must keep one rounding per written operation.
 */
#include "prl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------------------------------------
 * Argument checks and geometry: binarizeSauvola.cpp:38-47 (checks), :57 (w), :65 (pad), :66 (rect
 * from the PADDED size: Sauvola, Niblack) vs binarizeWolfJolion.cpp:69 / binarizeNICK.cpp:69 /
 * binarizeFeng.cpp:66 (rect from the UNPADDED size, computed before copyMakeBorder).
 * ------------------------------------------------------------------------------------------- */
int prl_oracle_binarize_geometry(const prl_binarize_params* p, int width, int height,
                                 prl_binarize_geometry* g)
{
    if (!p || !g) return PRL_ERR_BAD_ARG;
    memset(g, 0, sizeof(*g));
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (!((p->window_size > 1) && ((p->window_size % 2) == 1))) return PRL_ERR_BAD_WINDOW;
    if (p->method < PRL_SAUVOLA || p->method > PRL_FENG) return PRL_ERR_BAD_ARG;

    const int w = imin(p->window_size, imin(width, height));
    g->w = w;
    g->half = w / 2;
    g->padded_w = width + 2 * g->half;
    g->padded_h = height + 2 * g->half;
    if (p->method == PRL_SAUVOLA || p->method == PRL_NIBLACK) {
        g->out_w = g->padded_w - w;
        g->out_h = g->padded_h - w;
    } else {
        g->out_w = width - w;
        g->out_h = height - w;
    }
    if (g->out_w <= 0 || g->out_h <= 0) return PRL_ERR_EMPTY_RECT;
    return PRL_OK;
}

/* cv::copyMakeBorder(in, in, h, h, h, h, BORDER_REPLICATE) — binarizeSauvola.cpp:65 */
void prl_oracle_pad_replicate(const uint8_t* src, size_t src_step, int width, int height, int half,
                              uint8_t* dst, size_t dst_step)
{
    const int pw = width + 2 * half, ph = height + 2 * half;
    for (int y = 0; y < ph; ++y) {
        const int sy = imin(imax(y - half, 0), height - 1);
        const uint8_t* s = src + (size_t)sy * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < pw; ++x) d[x] = s[imin(imax(x - half, 0), width - 1)];
    }
}

/* cv::integral(in, sum, sqsum, CV_64F) then Rect(1,1,...) crop — binarizeSauvola.cpp:72-77.
 * [upstream] integral_ keeps a per-row running sum and adds the row above; every value is an
 * integer below 2^53, so any summation order gives the same doubles. */
void prl_oracle_integrals(const uint8_t* padded, size_t step, int pw, int ph, double* ii, double* iq)
{
    for (int y = 0; y < ph; ++y) {
        const uint8_t* s = padded + (size_t)y * step;
        double* r = ii + (size_t)y * pw;
        double* q = iq + (size_t)y * pw;
        const double* rp = y ? r - pw : NULL;
        const double* qp = y ? q - pw : NULL;
        double rs = 0.0, qs = 0.0;
        for (int x = 0; x < pw; ++x) {
            const double v = (double)s[x];
            rs += v;
            qs += v * v;
            r[x] = rs + (rp ? rp[x] : 0.0);
            q[x] = qs + (qp ? qp[x] : 0.0);
        }
    }
}

/* cv::filter2D with the 4-nonzero w x w kernel, anchor (-1,-1) = centre (w/2,w/2), on the ROI
 * starting at (half,half) of the cropped integral — binarizeSauvola.cpp:83-90.
 * [upstream] Filter2D<double,...> visits the non-zero taps in row-major order
 *   (0,0)=+f  (0,w-1)=-f  (w-1,0)=-f  (w-1,w-1)=+f   and does  s0 = delta; s0 += kf[k]*src_k.
 * ROI offset and anchor cancel, so output (y,x) reads integral (y+r, x+c) for tap (r,c). */
static inline double box4(const double* t, size_t stride, int y, int x, int w, double f)
{
    const double nf = -f;
    const double* r0 = t + (size_t)y * stride + x;
    const double* r1 = t + (size_t)(y + w - 1) * stride + x;
    double s0 = 0.0;
    s0 += f * r0[0];
    s0 += nf * r0[w - 1];
    s0 += nf * r1[0];
    s0 += f * r1[w - 1];
    return s0;
}

/* saturate_cast<uchar>(double): cvRound (SSE2 cvtsd2si, round-half-even; NaN, +-inf and anything
 * that does not fit int32 give INT_MIN) then clamp — the convertTo(CV_8UC1) at binarizeSauvola.cpp:119 */
uint8_t prl_oracle_sat_u8(double v)
{
    if (v != v) return 0;
    const double r = nearbyint(v); /* default rounding mode: to nearest, ties to even */
    if (!(r >= -2147483648.0 && r <= 2147483647.0)) return 0; /* "integer indefinite" = INT_MIN -> 0 */
    const long iv = (long)r;
    return (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

typedef struct plane_ctx {
    prl_binarize_geometry g;
    uint8_t* padded;
    double* ii;
    double* iq;
    double f;
} plane_ctx;

static void ctx_free(plane_ctx* c)
{
    free(c->padded);
    free(c->ii);
    free(c->iq);
    memset(c, 0, sizeof(*c));
}

static int ctx_build(plane_ctx* c, const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                     int width, int height)
{
    memset(c, 0, sizeof(*c));
    if (!src) return PRL_ERR_BAD_ARG;
    int st = prl_oracle_binarize_geometry(p, width, height, &c->g);
    if (st != PRL_OK) return st;
    if (src_step < (size_t)width) return PRL_ERR_BAD_ARG;
    const size_t pw = (size_t)c->g.padded_w, ph = (size_t)c->g.padded_h;
    c->padded = (uint8_t*)malloc(pw * ph);
    c->ii = (double*)malloc(pw * ph * sizeof(double));
    c->iq = (double*)malloc(pw * ph * sizeof(double));
    if (!c->padded || !c->ii || !c->iq) {
        ctx_free(c);
        return PRL_ERR_NOMEM;
    }
    prl_oracle_pad_replicate(src, src_step, width, height, c->g.half, c->padded, pw);
    prl_oracle_integrals(c->padded, pw, (int)pw, (int)ph, c->ii, c->iq);
    /* int wSqr = w*w; double wSqrBack = 1.0 / static_cast<double>(wSqr);  binarizeSauvola.cpp:58-59 */
    c->f = 1.0 / (double)(c->g.w * c->g.w);
    return PRL_OK;
}

/* m, s at one output position — binarizeSauvola.cpp:89-110 (identical text in all five files):
 *   m  = filter2D(II)            :89-90
 *   m2 = m.mul(m)                :93
 *   q  = filter2D(IIsq)          :106-107
 *   q -= m2 ; sqrt(q)            :109-110   (negative -> NaN, [upstream] cv::sqrt = IEEE sqrt) */
static inline void mean_dev_at(const plane_ctx* c, int y, int x, double* m, double* s)
{
    const size_t stride = (size_t)c->g.padded_w;
    const double mm = box4(c->ii, stride, y, x, c->g.w, c->f);
    const double m2 = mm * mm;
    double q = box4(c->iq, stride, y, x, c->g.w, c->f);
    q = q - m2;
    *m = mm;
    *s = sqrt(q);
}

int prl_oracle_mean_dev(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                        int width, int height, double* mean, double* dev)
{
    plane_ctx c;
    int st = ctx_build(&c, p, src, src_step, width, height);
    if (st != PRL_OK) return st;
    for (int y = 0; y < c.g.out_h; ++y)
        for (int x = 0; x < c.g.out_w; ++x) {
            double m, s;
            mean_dev_at(&c, y, x, &m, &s);
            if (mean) mean[(size_t)y * c.g.out_w + x] = m;
            if (dev) dev[(size_t)y * c.g.out_w + x] = s;
        }
    ctx_free(&c);
    return PRL_OK;
}

/* Global statistics used by Wolf-Jolion (and computed-but-unused by NICK, binarizeNICK.cpp:115-119).
 * [upstream] cv::minMaxLoc on float64 compares with < and >, so a NaN never becomes min or max;
 * the running max starts at -DBL_MAX. */
static void global_stats(const plane_ctx* c, double* image_min, double* dev_max)
{
    /* cv::minMaxLoc(imageInput, &imageMin) on the padded image — binarizeWolfJolion.cpp:115-116 */
    int mn = 255;
    const size_t n = (size_t)c->g.padded_w * c->g.padded_h;
    for (size_t i = 0; i < n; ++i)
        if (c->padded[i] < mn) mn = c->padded[i];
    *image_min = (double)mn;
    if (!dev_max) return;
    /* cv::minMaxLoc(localDevianceValues, &devianceMin, &devianceMax) — binarizeWolfJolion.cpp:118-119 */
    double mx = -DBL_MAX;
    for (int y = 0; y < c->g.out_h; ++y)
        for (int x = 0; x < c->g.out_w; ++x) {
            double m, s;
            mean_dev_at(c, y, x, &m, &s);
            if (s > mx) mx = s;
        }
    *dev_max = mx;
}

typedef struct thr_consts {
    int method;
    double k;
    double a, b;          /* Sauvola: k*RBack, 1-k */
    double imin, coeff;   /* Wolf: imageMin, k/devianceMax ; Feng: imageMin */
    double alpha1, k1, k2, gamma;
} thr_consts;

static void thr_prepare(thr_consts* t, const prl_binarize_params* p, const plane_ctx* c)
{
    memset(t, 0, sizeof(*t));
    t->method = p->method;
    t->k = p->k;
    switch (p->method) {
    case PRL_SAUVOLA: {
        /* const double R = 128; const double RBack = 1.0 / R;  binarizeSauvola.cpp:61-62 */
        const double R = 128;
        const double RBack = 1.0 / R;
        t->a = (p->k * RBack); /* convertTo alpha, :117 */
        t->b = (1.0 - p->k);   /* convertTo beta,  :117 */
        break;
    }
    case PRL_WOLFJOLION: {
        double dmax;
        global_stats(c, &t->imin, &dmax);
        t->coeff = p->k / dmax; /* binarizeWolfJolion.cpp:121 */
        break;
    }
    case PRL_FENG:
        global_stats(c, &t->imin, NULL); /* binarizeFeng.cpp:111-112 */
        t->alpha1 = p->feng_alpha1;
        t->k1 = p->feng_k1;
        t->k2 = p->feng_k2;
        t->gamma = p->feng_gamma;
        break;
    default:
        break;
    }
}

/* [upstream] cv::pow on float64 for the only inputs Feng ever feeds it (0, 1, NaN). power==2 is
 * multiply(src,src); other powers of 0/1 are 0/1 for gamma>0 and 1 for gamma==0. */
static inline double feng_pow(double r, double gamma)
{
    if (gamma == 2.0) return r * r;
    return pow(r, gamma);
}

/* Threshold value (float64, before the u8 cast) from m and s. */
static inline double threshold_from(const thr_consts* t, double m, double s)
{
    switch (t->method) {
    case PRL_SAUVOLA: {
        /* s.convertTo(s, f64, k*RBack, 1-k): s*alpha + beta  binarizeSauvola.cpp:115-117
         * T = m.mul(s)                                         :118 */
        const double d = s * t->a + t->b;
        return m * d;
    }
    case PRL_NIBLACK:
        /* localMeanValues + k * localDevianceValues  binarizeNiblack.cpp:108
         * [upstream] MatExpr A + k*B lowers to scaleAdd(B, k, A): B*k + A */
        return s * t->k + m;
    case PRL_WOLFJOLION: {
        /* s.convertTo(s, f64, coeff, -k)            binarizeWolfJolion.cpp:128
         * s = s.mul(m - imageMin)                   :129  ([upstream] m*1.0 + (-imageMin))
         * T = m + s                                 :130 */
        const double d = s * t->coeff + (-t->k);
        const double e = m * 1.0 + (-t->imin);
        const double g = d * e;
        return m + g;
    }
    case PRL_NICK: {
        /* C = m.mul(m); s = s.mul(s); C = C + s; sqrt(C)   binarizeNICK.cpp:121-124
         * addWeighted(m, 1, C, k, 0, T)                    :126  ((m*1 + C*k) + 0) */
        double C = m * m;
        const double s2 = s * s;
        C = C + s2;
        C = sqrt(C);
        return (m * 1.0 + C * t->k) + 0.0;
    }
    case PRL_FENG: {
        /* Rs = localDevianceValues (alias)                          binarizeFeng.cpp:118
         * divide(s, Rs, tmpAlpha1)  [upstream <=3.x: x/0 -> 0]      :124
         * pow(tmpAlpha1, gamma, tmpAlpha2)                          :126
         * alpha2 = k1*tmpAlpha2 (never used)                        :128
         * alpha3 = k2*tmpAlpha2  ([upstream] r2*k2 + 0)             :129
         * c1 = 1 - alpha1 ; c2 = tmpAlpha2.mul(tmpAlpha1)           :133-134
         * addWeighted(alpha3, imageMin, c2, -imageMin, 0, c3)       :137
         * T = c2 + c1 ; T = T.mul(m) ; T += c3                      :140-142 */
        const double r = (s != 0.0) ? (s / s) : 0.0;
        const double r2 = feng_pow(r, t->gamma);
        const double a3 = r2 * t->k2 + 0.0;
        const double c1 = 1.0 - t->alpha1;
        const double c2 = r2 * r;
        const double c3 = (a3 * t->imin + c2 * (-t->imin)) + 0.0;
        double T = c2 * 1.0 + c1;
        T = T * m;
        T = T + c3;
        return T;
    }
    default:
        return NAN;
    }
}

int prl_oracle_threshold_plane(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                               int width, int height, double* T)
{
    plane_ctx c;
    int st = ctx_build(&c, p, src, src_step, width, height);
    if (st != PRL_OK) return st;
    thr_consts t;
    thr_prepare(&t, p, &c);
    for (int y = 0; y < c.g.out_h; ++y)
        for (int x = 0; x < c.g.out_w; ++x) {
            double m, s;
            mean_dev_at(&c, y, x, &m, &s);
            T[(size_t)y * c.g.out_w + x] = threshold_from(&t, m, s);
        }
    ctx_free(&c);
    return PRL_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Morphology: cv::dilate / cv::erode(out, out, Mat(), Point(-1,-1), n) — binarizeSauvola.cpp:125-134.
 * [upstream] an empty kernel is the 3x3 rectangle; n iterations of a full rectangle are folded into
 * one (2n+1)x(2n+1) rectangle; the default border value makes out-of-image pixels neutral.
 * ------------------------------------------------------------------------------------------- */
static void rect_minmax(const uint8_t* src, size_t src_step, int width, int height, int n, int take_max,
                        uint8_t* dst, size_t dst_step, uint8_t* tmp)
{
    /* rows, then columns: max/min over a rectangle is separable */
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* t = tmp + (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            const int x0 = imax(x - n, 0), x1 = imin(x + n, width - 1);
            uint8_t v = s[x0];
            for (int j = x0 + 1; j <= x1; ++j)
                v = take_max ? (s[j] > v ? s[j] : v) : (s[j] < v ? s[j] : v);
            t[x] = v;
        }
    }
    for (int y = 0; y < height; ++y) {
        const int y0 = imax(y - n, 0), y1 = imin(y + n, height - 1);
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x) {
            uint8_t v = tmp[(size_t)y0 * width + x];
            for (int i = y0 + 1; i <= y1; ++i) {
                const uint8_t u = tmp[(size_t)i * width + x];
                v = take_max ? (u > v ? u : v) : (u < v ? u : v);
            }
            d[x] = v;
        }
    }
}

void prl_oracle_morph(int morph_iterations, const uint8_t* src, size_t src_step, int width, int height,
                      uint8_t* dst, size_t dst_step)
{
    if (morph_iterations == 0) {
        for (int y = 0; y < height; ++y)
            memmove(dst + (size_t)y * dst_step, src + (size_t)y * src_step, (size_t)width);
        return;
    }
    const int n = morph_iterations > 0 ? morph_iterations : -morph_iterations;
    uint8_t* tmp = (uint8_t*)malloc((size_t)width * height);
    uint8_t* mid = (uint8_t*)malloc((size_t)width * height);
    /* n > 0: dilate then erode (:127-128); n < 0: erode then dilate (:132-133) */
    rect_minmax(src, src_step, width, height, n, morph_iterations > 0, mid, (size_t)width, tmp);
    rect_minmax(mid, (size_t)width, width, height, n, morph_iterations < 0, dst, dst_step, tmp);
    free(tmp);
    free(mid);
}

/* ---------------------------------------------------------------------------------------------
 * The function body after cvtColor, all five methods.
 * ------------------------------------------------------------------------------------------- */
int prl_oracle_binarize(const prl_binarize_params* p, const uint8_t* src, size_t src_step,
                        int width, int height, uint8_t* dst, size_t dst_step)
{
    plane_ctx c;
    int st = ctx_build(&c, p, src, src_step, width, height);
    if (st != PRL_OK) return st;
    if (!dst || dst_step < (size_t)c.g.out_w) {
        ctx_free(&c);
        return PRL_ERR_BAD_ARG;
    }
    thr_consts t;
    thr_prepare(&t, p, &c);

    const int ow = c.g.out_w, oh = c.g.out_h, half = c.g.half;
    const int morph = p->morph_iterations;
    uint8_t* raw = dst;
    size_t raw_step = dst_step;
    if (morph != 0) {
        raw = (uint8_t*)malloc((size_t)ow * oh);
        raw_step = (size_t)ow;
        if (!raw) {
            ctx_free(&c);
            return PRL_ERR_NOMEM;
        }
    }
    for (int y = 0; y < oh; ++y) {
        /* imageInput(processingRect): the padded image at (half+y, half+x) — binarizeSauvola.cpp:122 */
        const uint8_t* prow = c.padded + (size_t)(y + half) * c.g.padded_w + half;
        uint8_t* o = raw + (size_t)y * raw_step;
        for (int x = 0; x < ow; ++x) {
            double m, s;
            mean_dev_at(&c, y, x, &m, &s);
            const double T = threshold_from(&t, m, s);
            const uint8_t t8 = prl_oracle_sat_u8(T);      /* convertTo(CV_8UC1)  :119 */
            o[x] = (prow[x] > t8) ? 255 : 0;              /* in(rect) > T        :122 */
        }
    }
    if (morph != 0) {
        prl_oracle_morph(morph, raw, raw_step, ow, oh, dst, dst_step);
        free(raw);
    }
    ctx_free(&c);
    return PRL_OK;
}

int prl_oracle_binarize_batch(const prl_binarize_params* p, int n_pages,
                              const uint8_t* src, size_t src_page_stride, size_t src_step,
                              int width, int height,
                              uint8_t* dst, size_t dst_page_stride, size_t dst_step, int threads)
{
    int status = PRL_OK;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int i = 0; i < n_pages; ++i) {
        int st = prl_oracle_binarize(p, src + (size_t)i * src_page_stride, src_step, width, height,
                                     dst + (size_t)i * dst_page_stride, dst_step);
        if (st != PRL_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            status = st;
        }
    }
    return status;
}

/* ---------------------------------------------------------------------------------------------
 * cv::cvtColor(in, in, COLOR_BGR2GRAY) on 8-bit — binarizeSauvola.cpp:51.
 * [upstream, version dependent] RGB2Gray<uchar>: yuv_shift = 14, B2Y=1868, G2Y=9617, R2Y=4899,
 *   gray = (b*B2Y + g*G2Y + r*R2Y + (1 << 13)) >> 14.     (tolerance +-1
 *   against OpenCV builds that use the 15-bit variant.)
 * ------------------------------------------------------------------------------------------- */
void prl_oracle_bgr2gray(const uint8_t* bgr, size_t src_step, int width, int height, int channels,
                         uint8_t* gray, size_t dst_step)
{
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = bgr + (size_t)y * src_step;
        uint8_t* d = gray + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x, s += channels)
            d[x] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
    }
}

/* ---------------------------------------------------------------------------------------------
 * cv::threshold(img, img, 128, 255, THRESH_BINARY | THRESH_OTSU) — the only Otsu in the reference
 * (deskew.cpp:224); BASELINE config 1 wraps it as prl::binarize.
 * [upstream] getThreshVal_Otsu_8u: 256-bin histogram, float64 between-class variance, first maximum.
 * ------------------------------------------------------------------------------------------- */
int prl_oracle_otsu(const uint8_t* src, size_t src_step, int width, int height,
                    uint8_t* dst, size_t dst_step)
{
    int h[256] = {0};
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        for (int x = 0; x < width; ++x) h[s[x]]++;
    }
    double mu = 0, scale = 1. / ((double)width * height);
    for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; ++i) {
        double p_i, q2, mu2, sigma;
        p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    const int thr = (int)max_val;
    if (dst)
        for (int y = 0; y < height; ++y) {
            const uint8_t* s = src + (size_t)y * src_step;
            uint8_t* d = dst + (size_t)y * dst_step;
            for (int x = 0; x < width; ++x) d[x] = s[x] > thr ? 255 : 0;
        }
    return thr;
}
