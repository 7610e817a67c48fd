/*
 * prl_oracle_bgnorm.c — CPU restatement of prl::backgroundNormalization.
 *
 * TEST INFRASTRUCTURE ONLY (see prl_oracle.h).  PARITY STATUS: **parity unpinned**.
 *
 * Reference: src/backgroundNormalization.cpp:36-61
 *     pixs = prl::opencvToLeptonica(&input);  pixn = pixBackgroundNormSimple(pixs, NULL, NULL);
 *     output = prl::leptonicaToOpenCV(pixn);
 * with the Mat <-> PIX conventions of src/formatConvert.cpp:38-109 (in) and :111-218 (out):
 *   1 channel  -> 8 bpp PIX, bytes copied                                   (formatConvert.cpp:68-76, :164-175)
 *   3 channels -> 32 bpp PIX, Mat byte 0/1/2 of a pixel -> the RED/GREEN/BLUE slots (:78-90), back as 3 channels from
 *                 the same slots (:193-206): a BGR Mat therefore travels with B in Leptonica's "red" slot; the only
 *                 channel that is singled out below is slot 1 (green in both orders)
 *   4 channels -> 32 bpp PIX, byte i -> bits 31-8i (:92-101); the result Mat has THREE channels (bytes 0..2), the
 *                 fourth input byte is dropped (:193-206)
 *
 * The arithmetic is Leptonica's, which is not in this image (no liblept, no headers; CMakeLists.txt:33 links -llept
 * unpinned).  What follows restates the published algorithm of leptonica 1.7x [upstream], function by function:
 *   adaptmap.c  pixBackgroundNormSimple  -> pixBackgroundNorm(pixs, NULL, NULL, sx=10, sy=15, thresh=60, mincount=40,
 *                                           bgval=200, smoothx=2, smoothy=1)
 *   adaptmap.c  pixGetBackgroundGrayMap / pixGetBackgroundRGBMap
 *                 fg mask  = pixThresholdToBinary(gray, 60)  (pixel < 60 -> 1), then pixMorphSequence("d7.1 + d1.7"):
 *                            dilation by a 7x1 and then a 1x7 brick, origin at the centre, pixels beyond the image = 0;
 *                            RGB: gray = pixConvertRGBToGrayFast = the GREEN slot
 *                 map      = (w+sx-1)/sx x (h+sy-1)/sy, 8 bpp, zero-initialised; only COMPLETE tiles (nx = w/sx,
 *                            ny = h/sy) are evaluated: sum and count of the pixels outside the fg mask, and if
 *                            count >= mincount the map pixel becomes sum / count (integer division)
 *   adaptmap.c  pixFillMapHoles(map, nx, ny, L_FILL_BLACK): per column j < nx: first non-zero row replicated upwards,
 *                 then zeros take the value above them, down to the LAST map row (including the row of incomplete
 *                 tiles); columns without data take their left neighbour (those left of the first good column its
 *                 value); if w_map > nx the last column is a copy of column w_map-2; no data in any column -> failure
 *   adaptmap.c  pixGetInvBackgroundMap(map, 200, 2, 1): maps narrower or lower than 5 fail; pixBlockconv(map, 2, 1);
 *                 16-bit inverse map val16 = (256*200)/smoothed (smoothed == 0 -> 200/2)
 *   convolve.c  pixBlockconv -> pixBlockconvGray -> blockconvLow: 32-bit inclusive accumulator a[i][j]; for every pixel
 *                 imin = max(i-1-hc,0), imax = min(i+hc,h-1), jmin/jmax likewise,
 *                 val = a[imax][jmax]-a[imax][jmin]-a[imin][jmax]+a[imin][jmin]  (so row/column 0 drop out of windows that
 *                 touch the top/left edge), (l_uint8)(norm*val + 0.5) with float32 norm = 1/(fwc*fhc); then the
 *                 boundary rows/columns are rescaled by float32 factors fhc/hn, fwc/wn, (l_uint8)min(val*normh*normw, 255)
 *   adaptmap.c  pixApplyInvBackgroundGrayMap / pixApplyInvBackgroundRGBMap: out = min(255, (p * val16[y/sy][x/sx]) / 256)
 * Failure of the map ("map not made", "pixmi not made") returns a copy of the source (newer leptonica; older versions
 * returned NULL for the second, which crashes the reference's converter - undefined there, a copy here).
 *
 * Everything is integer except blockconvLow's float32 scaling, restated with one float32 rounding per operation the
 * way an x86-64 SSE build evaluates it (FLT_EVAL_METHOD 0, no FMA).
 */
#include <stdlib.h>
#include <string.h>

#include "prl_oracle.h"

enum { SX = 10, SY = 15, THRESH = 60, MINCOUNT = 40, BGVAL = 200, SMOOTHX = 2, SMOOTHY = 1 };

void prl_oracle_bgnorm_map_size(int width, int height, int* map_w, int* map_h)
{
    *map_w = (width + SX - 1) / SX;
    *map_h = (height + SY - 1) / SY;
}

/* pixThresholdToBinary(gray, 60) + pixMorphSequence("d7.1 + d1.7"): fg[y*width+x] in {0,1}. */
void prl_oracle_bgnorm_fgmask(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* fg)
{
    const int gch = channels == 1 ? 0 : 1; /* pixConvertRGBToGrayFast: the green slot */
    uint8_t* b = (uint8_t*)malloc((size_t)width * height);
    uint8_t* hd = (uint8_t*)malloc((size_t)width * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) b[(size_t)y * width + x] = src[(size_t)y * src_step + (size_t)x * channels + gch] < THRESH;
    for (int y = 0; y < height; ++y) /* d7.1: horizontal brick of 7, origin 3 */
        for (int x = 0; x < width; ++x) {
            uint8_t v = 0;
            for (int d = -3; d <= 3; ++d)
                if (x + d >= 0 && x + d < width) v |= b[(size_t)y * width + x + d];
            hd[(size_t)y * width + x] = v;
        }
    for (int y = 0; y < height; ++y) /* d1.7: vertical brick of 7 */
        for (int x = 0; x < width; ++x) {
            uint8_t v = 0;
            for (int d = -3; d <= 3; ++d)
                if (y + d >= 0 && y + d < height) v |= hd[(size_t)(y + d) * width + x];
            fg[(size_t)y * width + x] = v;
        }
    free(b);
    free(hd);
}

/* pixFillMapHoles(pix, nx, ny, L_FILL_BLACK) on a w x h byte map; returns 1 when no column has data. */
static int fill_map_holes(uint8_t* m, int w, int h, int nx, int ny)
{
    uint8_t* na = (uint8_t*)calloc((size_t)nx + 1, 1);
    int nmiss = 0;
    for (int j = 0; j < nx; ++j) {
        int found = 0, y = 0;
        uint8_t val = 0;
        for (int i = 0; i < ny; ++i) {
            val = m[(size_t)i * w + j];
            if (val != 0) { y = i; found = 1; break; }
        }
        if (!found) { na[j] = 0; ++nmiss; continue; }
        na[j] = 1;
        for (int i = y - 1; i >= 0; --i) m[(size_t)i * w + j] = val;
        uint8_t lastval = m[j];
        for (int i = 1; i < h; ++i) {
            val = m[(size_t)i * w + j];
            if (val == 0) m[(size_t)i * w + j] = lastval;
            else lastval = val;
        }
    }
    na[nx] = 0; /* "last column" */
    if (nmiss == nx) { free(na); return 1; }
    if (nmiss > 0) {
        int goodcol = 0;
        for (int j = 0; j < w; ++j)
            if (j <= nx && na[j] == 1) { goodcol = j; break; }
        for (int j = goodcol - 1; j >= 0; --j)
            for (int i = 0; i < h; ++i) m[(size_t)i * w + j] = m[(size_t)i * w + j + 1];
        for (int j = goodcol + 1; j < w; ++j)
            if (j <= nx && na[j] == 0)
                for (int i = 0; i < h; ++i) m[(size_t)i * w + j] = m[(size_t)i * w + j - 1];
    }
    if (w > nx)
        for (int i = 0; i < h; ++i) m[(size_t)i * w + w - 1] = m[(size_t)i * w + w - 2];
    free(na);
    return 0;
}

/* pixGetBackgroundGrayMap / ...RGBMap after the fg mask: tile averages + hole filling for channel `c`.
 * Returns 1 if the map could not be made. */
int prl_oracle_bgnorm_bgmap(int channels, int c, const uint8_t* src, size_t src_step, int width, int height,
                            const uint8_t* fg, uint8_t* map)
{
    int mw, mh;
    prl_oracle_bgnorm_map_size(width, height, &mw, &mh);
    const int nx = width / SX, ny = height / SY;
    memset(map, 0, (size_t)mw * mh);
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j) {
            int sum = 0, count = 0;
            for (int k = 0; k < SY; ++k)
                for (int m = 0; m < SX; ++m) {
                    const int y = i * SY + k, x = j * SX + m;
                    if (fg[(size_t)y * width + x] == 0) {
                        sum += src[(size_t)y * src_step + (size_t)x * channels + c];
                        ++count;
                    }
                }
            if (count >= MINCOUNT) map[(size_t)i * mw + j] = (uint8_t)(sum / count);
        }
    return fill_map_holes(map, mw, mh, nx, ny);
}

/* pixBlockconv(map, wc=2, hc=1) for a map of at least 5 x 5 (blockconvLow). */
void prl_oracle_bgnorm_blockconv(const uint8_t* map, int w, int h, uint8_t* out)
{
    const int wc = SMOOTHX, hc = SMOOTHY;
    uint32_t* a = (uint32_t*)malloc((size_t)w * h * sizeof(uint32_t));
    for (int i = 0; i < h; ++i) { /* pixBlockconvAccum: inclusive 2-D prefix sums */
        uint32_t row = 0;
        for (int j = 0; j < w; ++j) {
            row += map[(size_t)i * w + j];
            a[(size_t)i * w + j] = row + (i > 0 ? a[(size_t)(i - 1) * w + j] : 0);
        }
    }
    const int fwc = 2 * wc + 1, fhc = 2 * hc + 1, wmwc = w - wc, hmhc = h - hc;
    const float norm = (float)(1.0 / ((float)fwc * fhc));
    for (int i = 0; i < h; ++i) {
        const int imin = i - 1 - hc > 0 ? i - 1 - hc : 0, imax = i + hc < h - 1 ? i + hc : h - 1;
        for (int j = 0; j < w; ++j) {
            const int jmin = j - 1 - wc > 0 ? j - 1 - wc : 0, jmax = j + wc < w - 1 ? j + wc : w - 1;
            uint32_t val = a[(size_t)imax * w + jmax] - a[(size_t)imax * w + jmin] + a[(size_t)imin * w + jmin] -
                           a[(size_t)imin * w + jmax];
            const float prod = norm * (float)val;
            out[(size_t)i * w + j] = (uint8_t)((double)prod + 0.5);
        }
    }
    free(a);
#define FIX2(i, j, nh, nw)                                              \
    do {                                                                \
        float t_ = (float)out[(size_t)(i) * w + (j)] * (nh);           \
        t_ = t_ * (nw);                                                 \
        out[(size_t)(i) * w + (j)] = (uint8_t)(t_ < 255.0f ? t_ : 255.0f); \
    } while (0)
#define FIX1(i, j, n1)                                                  \
    do {                                                                \
        float t_ = (float)out[(size_t)(i) * w + (j)] * (n1);           \
        out[(size_t)(i) * w + (j)] = (uint8_t)(t_ < 255.0f ? t_ : 255.0f); \
    } while (0)
    for (int i = 0; i <= hc; ++i) { /* first hc + 1 lines */
        const int hn = hc + i > 1 ? hc + i : 1;
        const float normh = (float)fhc / (float)hn;
        for (int j = 0; j <= wc; ++j) {
            const int wn = wc + j > 1 ? wc + j : 1;
            FIX2(i, j, normh, (float)fwc / (float)wn);
        }
        for (int j = wc + 1; j < wmwc; ++j) FIX1(i, j, normh);
        for (int j = wmwc; j < w; ++j) FIX2(i, j, normh, (float)fwc / (float)(wc + w - j));
    }
    for (int i = hmhc; i < h; ++i) { /* last hc lines */
        const float normh = (float)fhc / (float)(hc + h - i);
        for (int j = 0; j <= wc; ++j) {
            const int wn = wc + j > 1 ? wc + j : 1;
            FIX2(i, j, normh, (float)fwc / (float)wn);
        }
        for (int j = wc + 1; j < wmwc; ++j) FIX1(i, j, normh);
        for (int j = wmwc; j < w; ++j) FIX2(i, j, normh, (float)fwc / (float)(wc + w - j));
    }
    for (int i = hc + 1; i < hmhc; ++i) { /* intermediate lines */
        for (int j = 0; j <= wc; ++j) {
            const int wn = wc + j > 1 ? wc + j : 1;
            FIX1(i, j, (float)fwc / (float)wn);
        }
        for (int j = wmwc; j < w; ++j) FIX1(i, j, (float)fwc / (float)(wc + w - j));
    }
#undef FIX1
#undef FIX2
}

/* pixGetInvBackgroundMap(map, 200, 2, 1): returns 1 when the map is smaller than 5 x 5. */
int prl_oracle_bgnorm_invmap(const uint8_t* map, int w, int h, uint16_t* inv)
{
    if (w < 5 || h < 5) return 1;
    uint8_t* sm = (uint8_t*)malloc((size_t)w * h);
    prl_oracle_bgnorm_blockconv(map, w, h, sm);
    for (size_t i = 0; i < (size_t)w * h; ++i) inv[i] = sm[i] > 0 ? (uint16_t)((256 * BGVAL) / sm[i]) : (uint16_t)(BGVAL / 2);
    free(sm);
    return 0;
}

int prl_oracle_bgnorm_out_channels(int channels) { return channels == 1 ? 1 : 3; }

int prl_oracle_bgnorm(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                      size_t dst_step)
{
    if (width <= 0 || height <= 0 || !src) return PRL_ERR_EMPTY; /* backgroundNormalization.cpp:40-43 */
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS; /* formatConvert.cpp:103-104 */
    const int och = prl_oracle_bgnorm_out_channels(channels);
    if (!dst || src_step < (size_t)width * channels || dst_step < (size_t)width * och) return PRL_ERR_BAD_ARG;
    int mw, mh;
    prl_oracle_bgnorm_map_size(width, height, &mw, &mh);
    uint8_t* fg = (uint8_t*)malloc((size_t)width * height);
    uint8_t* map = (uint8_t*)malloc((size_t)mw * mh * 3);
    uint16_t* inv = (uint16_t*)malloc((size_t)mw * mh * 3 * sizeof(uint16_t));
    prl_oracle_bgnorm_fgmask(channels, src, src_step, width, height, fg);
    int failed = 0;
    for (int c = 0; c < och && !failed; ++c)
        failed = prl_oracle_bgnorm_bgmap(channels, c, src, src_step, width, height, fg, map + (size_t)c * mw * mh);
    for (int c = 0; c < och && !failed; ++c)
        failed = prl_oracle_bgnorm_invmap(map + (size_t)c * mw * mh, mw, mh, inv + (size_t)c * mw * mh);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < och; ++c) {
                const int p = src[(size_t)y * src_step + (size_t)x * channels + c];
                int v = p; /* failure: pixCopy(NULL, pixs) */
                if (!failed) {
                    v = (p * (int)inv[(size_t)c * mw * mh + (size_t)(y / SY) * mw + x / SX]) / 256;
                    if (v > 255) v = 255;
                }
                dst[(size_t)y * dst_step + (size_t)x * och + c] = (uint8_t)v;
            }
    free(fg);
    free(map);
    free(inv);
    return PRL_OK;
}
