/*
 * prl_oracle_deskew.c — CPU restatement of prl::deskew and prl::rotate.
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  PARITY STATUS: **parity unpinned** (no OpenCV in this image).
 *
 * Reference:
 *   prl::deskew       src/deskew/deskew.cpp:208-251   gray (:214-221) -> cv::threshold(128, 255, BINARY|OTSU) (:224) ->
 *                                                      findAngle (:226) -> prl::rotate(input, output, angle) (:230) ->
 *                                                      findOrientation (:238)
 *   prl::findAngle    src/deskew/deskew.cpp:139-205   bitwise_not, cv::HoughLinesP(input, lines, 1, CV_PI/180, 100,
 *                                                      width/8.f, 20) (:148), atan2 of every segment (:167-169), first-fit
 *                                                      clustering with eq_d(.., .., 0.01) (:172-193), most populated
 *                                                      cluster's first angle, in degrees (:195-201)
 *   prl::rotate       src/rotate.cpp:35-72            fmod(angle, 360); 90 / 180 / 270 by transpose + flip (:38-58), else
 *                                                      bitwise_not, getRotationMatrix2D((len/2, len/2), angle, 1),
 *                                                      warpAffine to len x len (len = max(cols, rows)), bitwise_not (:61-70)
 *   prl::findOrientation  src/deskew/deskew.cpp:70-136  needs Leptonica's pixOrientDetectDwa; prl::deskew always hands it
 *                          the 1-channel thresholded page, for which :73-76 leaves grayImage EMPTY, so the function runs
 *                          adaptiveThreshold on an empty Mat and converts an empty Mat to a NULL PIX: it either throws or
 *                          returns 0 (:81-84).  Canonical choice here: 0, i.e. the orientation step of prl::deskew is a no-op
 *                          (SURVEY.md Appendix D.9).  The second rotate (:231) only feeds that step and is skipped.
 *
 * OpenCV arithmetic restated [upstream, from OpenCV 3.x/4.x modules/imgproc/src/hough.cpp, imgwarp.cpp, core/rng]:
 *   HoughLinesProbabilistic: trig table (float)(cos((double)n*theta)*irho); points collected in raster order; cv::RNG
 *     seeded with (uint64)-1, next() = (unsigned)(state = (uint64)(unsigned)state*4164903690U + (state >> 32)),
 *     uniform(0,count) = next() % count; per point: skip if its mask byte is already cleared, vote r = cvRound(j*cos+i*sin)
 *     + (numrho-1)/2 for all 180 angles keeping the FIRST angle with the largest count above threshold-1; walk both ways in
 *     16.16 fixed point until the border or a gap above lineGap; a line is good if |dx| or |dy| >= lineLength; second walk
 *     clears the mask and, for a good line, takes the votes of the cleared points back.
 *   getRotationMatrix2D (double cos/sin of angle*CV_PI/180, Point2f centre), warpAffine's inversion of the matrix,
 *     adelta/bdelta = cvRound(M*x*1024), X0 = cvRound((M1*y+M2)*1024) + 16, X = (X0+adelta) >> 5, integer part
 *     saturate_cast<short>(X >> 5), fraction X & 31; remapBilinear with the 32x32 fixed-point weight table
 *     (32*(32-fx)*(32-fy) ... of 32768; entry (0,0) saturates to 32767 and OpenCV's repair puts the missing 1 on the
 *     diagonal neighbour - no effect on 8-bit results), (sum + 16384) >> 15, BORDER_CONSTANT value 0.
 *   One rounding per written floating-point operation, no FMA; cos/sin/atan2 are the host libm's, as in the reference.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "prl_oracle.h"

#ifndef M_PI
#define M_PI 3.1415926535897932384626433832795
#endif
#define CV_PI_ 3.1415926535897932384626433832795

static int cv_round(double v) { return (int)lrint(v); } /* round-half-even (default rounding mode) */

/* ---- cv::HoughLinesP ------------------------------------------------------------------------------------------- */

int prl_oracle_houghp(const uint8_t* image, size_t step, int width, int height, int threshold, int lineLength, int lineGap,
                      int32_t* lines, int cap)
{
    const float rho = 1.f, theta = (float)(CV_PI_ / 180);
    const float irho = 1 / rho;
    uint64_t rng = (uint64_t)-1;
    const int numangle = cv_round(CV_PI_ / theta);
    const int numrho = cv_round(((width + height) * 2 + 1) / rho);
    int* accum = (int*)calloc((size_t)numangle * numrho, sizeof(int));
    uint8_t* mask = (uint8_t*)malloc((size_t)width * height);
    float* ttab = (float*)malloc(sizeof(float) * 2 * numangle);
    int32_t* nz = (int32_t*)malloc(sizeof(int32_t) * 2 * ((size_t)width * height + 1));
    int n_lines = 0;
    for (int n = 0; n < numangle; n++) {
        ttab[n * 2] = (float)(cos((double)n * theta) * irho);
        ttab[n * 2 + 1] = (float)(sin((double)n * theta) * irho);
    }
    int count = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            if (image[(size_t)y * step + x]) {
                mask[(size_t)y * width + x] = 1;
                nz[2 * count] = x;
                nz[2 * count + 1] = y;
                count++;
            } else
                mask[(size_t)y * width + x] = 0;
        }
    for (; count > 0; count--) {
        rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        const int idx = (int)((unsigned)rng % (unsigned)count);
        int max_val = threshold - 1, max_n = 0;
        const int j = nz[2 * idx], i = nz[2 * idx + 1];
        int line_end[2][2] = {{0, 0}, {0, 0}}; /* [k] = {x, y} */
        int x0, y0, dx0, dy0, xflag;
        const int shift = 16;
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(size_t)i * width + j]) continue;
        for (int n = 0; n < numangle; n++) {
            const float fr = (float)j * ttab[n * 2] + (float)i * ttab[n * 2 + 1];
            int r = cv_round(fr);
            r += (numrho - 1) / 2;
            const int val = ++accum[(size_t)n * numrho + r];
            if (max_val < val) {
                max_val = val;
                max_n = n;
            }
        }
        if (max_val < threshold) continue;
        const float a = -ttab[max_n * 2 + 1], b = ttab[max_n * 2];
        x0 = j;
        y0 = i;
        if (fabs(a) > fabs(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = cv_round(b * (1 << shift) / fabs(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = cv_round(a * (1 << shift) / fabs(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        for (int k = 0; k < 2; k++) {
            int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; }
                else { j1 = x >> shift; i1 = y; }
                if (j1 < 0 || j1 >= width || i1 < 0 || i1 >= height) break;
                if (mask[(size_t)i1 * width + j1]) {
                    gap = 0;
                    line_end[k][1] = i1;
                    line_end[k][0] = j1;
                } else if (++gap > lineGap)
                    break;
            }
        }
        const int good_line = abs(line_end[1][0] - line_end[0][0]) >= lineLength || abs(line_end[1][1] - line_end[0][1]) >= lineLength;
        for (int k = 0; k < 2; k++) {
            int x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; }
                else { j1 = x >> shift; i1 = y; }
                uint8_t* m = mask + (size_t)i1 * width + j1;
                if (*m) {
                    if (good_line)
                        for (int n = 0; n < numangle; n++) {
                            const float fr = (float)j1 * ttab[n * 2] + (float)i1 * ttab[n * 2 + 1];
                            int r = cv_round(fr);
                            r += (numrho - 1) / 2;
                            accum[(size_t)n * numrho + r]--;
                        }
                    *m = 0;
                }
                if (i1 == line_end[k][1] && j1 == line_end[k][0]) break;
            }
        }
        if (good_line) {
            if (n_lines < cap) {
                lines[4 * n_lines] = line_end[0][0];
                lines[4 * n_lines + 1] = line_end[0][1];
                lines[4 * n_lines + 2] = line_end[1][0];
                lines[4 * n_lines + 3] = line_end[1][1];
            }
            n_lines++;
        }
    }
    free(accum);
    free(mask);
    free(ttab);
    free(nz);
    return n_lines;
}

/* The vote of deskew.cpp:158-201 over a list of segments (x0,y0,x1,y1): angle in degrees, 0 for an empty list. */
double prl_oracle_vote_angle(const int32_t* lines, int nb_lines)
{
    if (!nb_lines) return 0.0;
    double* first = (double*)malloc(sizeof(double) * nb_lines);
    int* second = (int*)malloc(sizeof(int) * nb_lines);
    int n_elem = 0;
    const double delta = 0.01;
    for (int l = 0; l < nb_lines; ++l) {
        const double ang = atan2((double)lines[4 * l + 3] - lines[4 * l + 1], (double)lines[4 * l + 2] - lines[4 * l]);
        int found = 0;
        for (int e = 0; e < n_elem; ++e)
            if (fabs(ang - first[e]) <= delta) {
                second[e]++;
                found = 1;
                break;
            }
        if (!found) {
            first[n_elem] = ang;
            second[n_elem] = 0;
            n_elem++;
        }
    }
    int best = 0; /* std::max_element: first of the largest */
    for (int e = 1; e < n_elem; ++e)
        if (second[best] < second[e]) best = e;
    const double r = first[best] * 180 / M_PI;
    free(first);
    free(second);
    return r;
}

/* prl::findAngle on the thresholded page (0 / 255). */
double prl_oracle_find_angle(const uint8_t* bin, size_t step, int width, int height, int* n_lines_out)
{
    uint8_t* inv = (uint8_t*)malloc((size_t)width * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) inv[(size_t)y * width + x] = (uint8_t)~bin[(size_t)y * step + x];
    int cap = 1 << 16;
    int32_t* lines = (int32_t*)malloc(sizeof(int32_t) * 4 * cap);
    int n = prl_oracle_houghp(inv, (size_t)width, width, height, 100, cv_round((double)(width / 8.f)), cv_round(20.0), lines, cap);
    if (n > cap) { /* rerun with room for everything */
        cap = n;
        lines = (int32_t*)realloc(lines, sizeof(int32_t) * 4 * cap);
        n = prl_oracle_houghp(inv, (size_t)width, width, height, 100, cv_round((double)(width / 8.f)), cv_round(20.0), lines, cap);
    }
    if (n_lines_out) *n_lines_out = n;
    const double ang = prl_oracle_vote_angle(lines, n);
    free(lines);
    free(inv);
    return ang;
}

/* ---- prl::rotate -------------------------------------------------------------------------------------------------- */

static int eq_d(double a, double b, double delta) { return fabs(a - b) <= delta; }

/* 0 = general (warpAffine), 1 = 90, 2 = 180, 3 = 270 */
int prl_oracle_rotate_kind(double angle)
{
    angle = fmod(angle, 360.0);
    if (eq_d(angle, 90.0, 1e-7)) return 1;
    if (eq_d(angle, 180.0, 1e-7)) return 2;
    if (eq_d(angle, 270.0, 1e-7)) return 3;
    return 0;
}

void prl_oracle_rotate_size(int width, int height, double angle, int* out_w, int* out_h)
{
    const int kind = prl_oracle_rotate_kind(angle);
    if (kind == 1 || kind == 3) { *out_w = height; *out_h = width; }
    else if (kind == 2) { *out_w = width; *out_h = height; }
    else { *out_w = *out_h = width > height ? width : height; }
}

/* The INVERTED 2x3 matrix warpAffine works with, for prl::rotate's general branch (rotate.cpp:64-68). */
void prl_oracle_rotate_matrix(int width, int height, double angle, double M[6])
{
    const int len = width > height ? width : height;
    const float cx = (float)(len / 2.0), cy = (float)(len / 2.0);
    angle = fmod(angle, 360.0);
    angle *= CV_PI_ / 180;
    const double alpha = cos(angle) * 1.0, beta = sin(angle) * 1.0;
    M[0] = alpha;
    M[1] = beta;
    M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta;
    M[4] = alpha;
    M[5] = beta * cx + (1 - alpha) * cy;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11;
    M[1] *= -D;
    M[3] *= -D;
    M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1;
    M[5] = b2;
}

static int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

int prl_oracle_rotate(int channels, const uint8_t* src, size_t src_step, int width, int height, double angle,
                      uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0 || !src || !dst) return PRL_ERR_EMPTY;
    if (channels < 1 || channels > 4) return PRL_ERR_BAD_CHANNELS;
    const int kind = prl_oracle_rotate_kind(angle), cn = channels;
    int ow, oh;
    prl_oracle_rotate_size(width, height, angle, &ow, &oh);
    if (kind != 0) {
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                int sx, sy;
                if (kind == 1) { sx = y; sy = height - 1 - x; }           /* transpose, flip around the y axis */
                else if (kind == 2) { sx = width - 1 - x; sy = height - 1 - y; }
                else { sx = width - 1 - y; sy = x; }                       /* transpose, flip around the x axis */
                memcpy(dst + (size_t)y * dst_step + (size_t)x * cn, src + (size_t)sy * src_step + (size_t)sx * cn, (size_t)cn);
            }
        return PRL_OK;
    }
    double M[6];
    prl_oracle_rotate_matrix(width, height, angle, M);
    for (int y = 0; y < oh; ++y) {
        const int X0 = cv_round((M[1] * y + M[2]) * 1024) + 16, Y0 = cv_round((M[4] * y + M[5]) * 1024) + 16;
        for (int x = 0; x < ow; ++x) {
            const int X = (X0 + cv_round(M[0] * x * 1024)) >> 5, Y = (Y0 + cv_round(M[3] * x * 1024)) >> 5;
            const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5), fx = X & 31, fy = Y & 31;
            const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
            for (int c = 0; c < cn; ++c) {
                int v[4];
                for (int t = 0; t < 4; ++t) {
                    const int px = sx + (t & 1), py = sy + (t >> 1);
                    /* the source of warpAffine is cv::bitwise_not(input); outside it the border value 0 */
                    v[t] = (px >= 0 && px < width && py >= 0 && py < height)
                               ? 255 - src[(size_t)py * src_step + (size_t)px * cn + c] : 0;
                }
                const int s = (v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15;
                dst[(size_t)y * dst_step + (size_t)x * cn + c] = (uint8_t)(255 - s); /* the final bitwise_not */
            }
        }
    }
    return PRL_OK;
}

/* prl::deskew: *out_w x *out_h pixels of `channels` channels into dst (room for max(W,H)^2 pixels).
 * angle_out / thr_out / n_lines_out (optional) report findAngle's result, the Otsu threshold and the segment count. */
int prl_oracle_deskew(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                      size_t dst_step, int* out_w, int* out_h, double* angle_out, int* thr_out, int* n_lines_out)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY; /* CV_Assert(!inputImage.empty()) */
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    uint8_t* gray = (uint8_t*)malloc((size_t)width * height);
    if (channels != 1) prl_oracle_bgr2gray(src, src_step, width, height, channels, gray, (size_t)width);
    else
        for (int y = 0; y < height; ++y) memcpy(gray + (size_t)y * width, src + (size_t)y * src_step, (size_t)width);
    const int thr = prl_oracle_otsu(gray, (size_t)width, width, height, gray, (size_t)width);
    if (thr_out) *thr_out = thr;
    const double angle = prl_oracle_find_angle(gray, (size_t)width, width, height, n_lines_out);
    free(gray);
    if (angle_out) *angle_out = angle;
    if (angle != 0 && isfinite(angle)) {
        prl_oracle_rotate_size(width, height, angle, out_w, out_h);
        if (dst_step < (size_t)*out_w * channels) return PRL_ERR_BAD_ARG;
        return prl_oracle_rotate(channels, src, src_step, width, height, angle, dst, dst_step);
    }
    *out_w = width;
    *out_h = height;
    if (dst_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, (size_t)width * channels);
    return PRL_OK;
}
