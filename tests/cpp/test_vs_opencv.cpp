// test_vs_opencv.cpp — SURVEY.md §8c last row: the oracle against REAL OpenCV (and Leptonica), where they exist.
//
// The reference cannot be built in the build image (no OpenCV, no Leptonica), so oracle/ restates what PRLib asks those
// libraries to compute and is "parity unpinned".  This program is the pin: compiled only against <opencv2/...> when the
// headers are found (the GPU box or a developer machine may have them), it issues the same OpenCV calls in the same
// order PRLib's hot path issues them (src/binarizations/binarizeSauvola.cpp:65-134 and the same lines of the other four
// binarizers, src/denoise/denoiseNLM.cpp:31, src/deskew/deskew.cpp:148,224, src/rotate.cpp:61-70,
// src/backgroundNormalization.cpp:36-61) - written here from SURVEY.md Appendix A, not taken from the PRLib sources, which
// never travel - and counts the pixels where the oracle's result differs.  It prints ONE JSON object; without OpenCV
// it prints {"opencv": null, ...} and exits 0, and the parity status stays "unpinned".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/prl_oracle.h"

#if !defined(PRL_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/imgproc.hpp>) && __has_include(<opencv2/photo.hpp>)
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/photo.hpp>
#define PRL_WITH_OPENCV 1
#endif
#if __has_include(<leptonica/allheaders.h>)
#include <leptonica/allheaders.h>
#define PRL_WITH_LEPTONICA 1
#endif
#endif

#ifndef PRL_WITH_OPENCV
int main()
{
#ifdef PRL_OPENCV_BUILD_FAILED
    std::printf("{\"opencv\": null, \"leptonica\": null, \"why\": \"build_failed\", \"note\": \"OpenCV was found but this program did not "
                "compile or link against it (tests/cpp/test_vs_opencv.build_error): oracle parity stays unpinned\"}\n");
#else
    std::printf("{\"opencv\": null, \"leptonica\": null, \"why\": \"headers_absent\", \"note\": \"OpenCV headers not found: oracle parity "
                "stays unpinned\"}\n");
#endif
    return 0;
}
#else

namespace {

// deterministic test pages: paper noise + dark strokes (same spirit as prlib_amd/synth.py, own LCG)
cv::Mat make_page(int rows, int cols, unsigned seed, int channels = 1)
{
    cv::Mat m(rows, cols, CV_MAKETYPE(CV_8U, channels));
    unsigned s = seed * 2654435761u + 97u;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            s = s * 1664525u + 1013904223u;
            int v = 205 + (int)((s >> 24) % 31) - 15;
            if (((x / 11) + (y / 4) * 3) % 9 == 0) v -= 110 + (int)((s >> 16) % 40);
            for (int c = 0; c < channels; ++c) {
                const int vc = v - 4 * c;
                m.ptr(y)[x * channels + c] = (unsigned char)(vc < 0 ? 0 : (vc > 255 ? 255 : vc));
            }
        }
    return m;
}

// Window mean and deviation planes the way all five binarizers build them: replicate border of w/2,
// float64 integrals without their zero row/column, the 4-tap +f -f -f +f kernel through cv::filter2D on the ROI.
struct Planes {
    cv::Mat mean, dev, padded;
    cv::Rect rect;
};

Planes local_planes(const cv::Mat& gray, int windowSize, bool rect_from_padded)
{
    Planes p;
    const int w = std::min(windowSize, std::min(gray.cols, gray.rows));
    const double f = 1.0 / static_cast<double>(w * w);
    const cv::Rect unpadded(w / 2, w / 2, gray.cols - w, gray.rows - w);
    cv::copyMakeBorder(gray, p.padded, w / 2, w / 2, w / 2, w / 2, cv::BORDER_REPLICATE);
    p.rect = rect_from_padded ? cv::Rect(w / 2, w / 2, p.padded.cols - w, p.padded.rows - w) : unpadded;
    cv::Mat ii, iq;
    cv::integral(p.padded, ii, iq, CV_64FC1);
    ii = ii(cv::Rect(1, 1, ii.cols - 1, ii.rows - 1));
    iq = iq(cv::Rect(1, 1, iq.cols - 1, iq.rows - 1));
    cv::Mat kernel = cv::Mat::zeros(w, w, CV_64FC1);
    kernel.at<double>(0, 0) = f;
    kernel.at<double>(w - 1, 0) = -f;
    kernel.at<double>(w - 1, w - 1) = f;
    kernel.at<double>(0, w - 1) = -f;
    cv::filter2D(ii(p.rect), p.mean, CV_64FC1, kernel, cv::Point(-1, -1), 0.0, cv::BORDER_REFLECT);
    cv::Mat mean2 = p.mean.mul(p.mean);
    cv::filter2D(iq(p.rect), p.dev, CV_64FC1, kernel);
    p.dev -= mean2;
    cv::sqrt(p.dev, p.dev);
    return p;
}

void close_or_open(cv::Mat& mask, int n)
{
    if (n > 0) {
        cv::dilate(mask, mask, cv::Mat(), cv::Point(-1, -1), n);
        cv::erode(mask, mask, cv::Mat(), cv::Point(-1, -1), n);
    } else if (n < 0) {
        cv::erode(mask, mask, cv::Mat(), cv::Point(-1, -1), -n);
        cv::dilate(mask, mask, cv::Mat(), cv::Point(-1, -1), -n);
    }
}

// The five binarizers through OpenCV, per SURVEY.md Appendix A.1-A.5.
cv::Mat binarize_opencv(const prl_binarize_params& bp, const cv::Mat& gray)
{
    const bool sn = bp.method == PRL_SAUVOLA || bp.method == PRL_NIBLACK;
    Planes p = local_planes(gray, bp.window_size, sn);
    cv::Mat T;
    const double k = bp.k;
    switch (bp.method) {
    case PRL_SAUVOLA: {
        cv::Mat d;
        p.dev.convertTo(d, p.dev.type(), k * (1.0 / 128.0), 1.0 - k);
        T = p.mean.mul(d);
        break;
    }
    case PRL_NIBLACK:
        T = p.mean + k * p.dev;
        break;
    case PRL_WOLFJOLION: {
        double imin = 0, smax = 0;
        cv::minMaxLoc(gray, &imin);
        cv::minMaxLoc(p.dev, nullptr, &smax);
        cv::Mat d, e;
        p.dev.convertTo(d, p.dev.type(), k / smax, -k);
        e = p.mean - imin;
        T = p.mean + d.mul(e);
        break;
    }
    case PRL_NICK: {
        cv::Mat c = p.mean.mul(p.mean);
        c = c + p.dev.mul(p.dev);
        cv::sqrt(c, c);
        cv::addWeighted(p.mean, 1.0, c, k, 0.0, T);
        break;
    }
    default: {  // Feng: Rs aliases the deviation plane, alpha2 / k1 are dead
        double imin = 0;
        cv::minMaxLoc(gray, &imin);
        cv::Mat r, r2, a3, c2, c3, t;
        cv::divide(p.dev, p.dev, r);
        cv::pow(r, bp.feng_gamma, r2);
        a3 = bp.feng_k2 * r2;
        c2 = r2.mul(r);
        cv::addWeighted(a3, imin, c2, -imin, 0.0, c3);
        t = c2 + (1.0 - bp.feng_alpha1);
        t = t.mul(p.mean);
        T = t + c3;
    }
    }
    cv::Mat T8;
    T.convertTo(T8, CV_8UC1);
    cv::Mat out = p.padded(p.rect) > T8;  // every method compares the PADDED page under its rectangle
    close_or_open(out, bp.morph_iterations);
    return out;
}

size_t count_diff(const cv::Mat& a, const unsigned char* b, size_t b_step)
{
    size_t bad = 0;
    const size_t row = (size_t)a.cols * a.channels();
    for (int y = 0; y < a.rows; ++y)
        for (size_t x = 0; x < row; ++x) bad += a.ptr(y)[x] != b[(size_t)y * b_step + x];
    return bad;
}

std::string binarizer_report(int method, int w, double k, int morph)
{
    prl_binarize_params bp{};
    bp.method = method; bp.window_size = w; bp.k = k; bp.morph_iterations = morph;
    bp.feng_alpha1 = 0.75; bp.feng_k1 = 0.2; bp.feng_k2 = 0.03; bp.feng_gamma = 2.0;
    size_t bad = 0, px = 0;
    for (unsigned seed = 1; seed <= 3; ++seed) {
        cv::Mat gray = make_page(300 + 17 * seed, 400 + 29 * seed, seed);
        prl_binarize_geometry g{};
        if (prl_oracle_binarize_geometry(&bp, gray.cols, gray.rows, &g) != PRL_OK) return "null";
        std::vector<unsigned char> want((size_t)g.out_w * g.out_h);
        prl_oracle_binarize(&bp, gray.data, gray.step, gray.cols, gray.rows, want.data(), (size_t)g.out_w);
        cv::Mat got = binarize_opencv(bp, gray);
        if (got.cols != g.out_w || got.rows != g.out_h) return "\"size mismatch\"";
        bad += count_diff(got, want.data(), (size_t)g.out_w);
        px += want.size();
    }
    char buf[96];
    std::snprintf(buf, sizeof buf, "{\"pixels\": %zu, \"mismatching\": %zu}", px, bad);
    return buf;
}

}  // namespace

int main()
{
    std::string js = "{\"opencv\": \"" CV_VERSION "\"";
    const char* names[5] = {"sauvola", "niblack", "wolfjolion", "nick", "feng"};
    const int wins[5] = {31, 101, 31, 21, 21};
    const double ks[5] = {0.34, 0.01, 0.3, -0.01, 0.0};
    const int morphs[5] = {0, 2, -1, 0, 2};
    for (int m = 0; m < 5; ++m) js += std::string(", \"") + names[m] + "\": " + binarizer_report(m, wins[m], ks[m], morphs[m]);
    {   // prl::denoise = cv::fastNlMeansDenoisingColored(src, dst, strength)
        cv::Mat bgr = make_page(96, 128, 7, 3), den;
        cv::fastNlMeansDenoisingColored(bgr, den, 10.0f);
        std::vector<unsigned char> want((size_t)96 * 128 * 3);
        prl_oracle_denoise(3, 10.0f, bgr.data, bgr.step, 128, 96, want.data(), 128 * 3, 4);
        size_t maxd = 0, bad = 0;
        for (int y = 0; y < 96; ++y)
            for (int x = 0; x < 128 * 3; ++x) {
                const int d = std::abs((int)den.ptr(y)[x] - (int)want[(size_t)y * 384 + x]);
                bad += d != 0;
                maxd = std::max<size_t>(maxd, (size_t)d);
            }
        char buf[96];
        std::snprintf(buf, sizeof buf, ", \"denoise\": {\"bytes\": %d, \"mismatching\": %zu, \"max_abs_diff\": %zu}", 96 * 128 * 3, bad, maxd);
        js += buf;
    }
    {   // Otsu + HoughLinesP + rotate (prl::deskew's pieces)
        cv::Mat gray = make_page(260, 340, 11), bin, inv;
        const double thr = cv::threshold(gray, bin, 128, 255, cv::THRESH_BINARY | cv::THRESH_OTSU);
        const int othr = prl_oracle_otsu(gray.data, gray.step, gray.cols, gray.rows, nullptr, 0);
        cv::bitwise_not(bin, inv);
        std::vector<cv::Vec4i> lines;
        cv::HoughLinesP(inv, lines, 1, CV_PI / 180, 60, gray.cols / 8.f, 20);
        std::vector<int32_t> ol(4 * 65536);
        const int n = prl_oracle_houghp(inv.data, inv.step, inv.cols, inv.rows, 60, cvRound(gray.cols / 8.f), 20, ol.data(), 65536);
        size_t ldiff = (size_t)std::abs(n - (int)lines.size());
        for (int i = 0; i < n && i < (int)lines.size(); ++i)
            for (int c = 0; c < 4; ++c) ldiff += lines[(size_t)i][c] != ol[(size_t)4 * i + c];
        cv::Mat src = make_page(120, 170, 13, 3), neg, rot;
        const int len = std::max(src.cols, src.rows);
        cv::bitwise_not(src, neg);
        cv::Mat r = cv::getRotationMatrix2D(cv::Point2f(static_cast<float>(len / 2.0), static_cast<float>(len / 2.0)), 3.7, 1.0);
        cv::warpAffine(neg, rot, r, cv::Size(len, len));
        cv::bitwise_not(rot, rot);
        std::vector<unsigned char> want((size_t)len * len * 3);
        prl_oracle_rotate(3, src.data, src.step, src.cols, src.rows, 3.7, want.data(), (size_t)len * 3);
        char buf[200];
        std::snprintf(buf, sizeof buf, ", \"otsu\": {\"opencv\": %d, \"oracle\": %d}, \"houghp\": {\"segments\": %zu, \"differences\": %zu}, "
                      "\"rotate\": {\"bytes\": %d, \"mismatching\": %zu}", (int)thr, othr, lines.size(), ldiff, len * len * 3,
                      count_diff(rot, want.data(), (size_t)len * 3));
        js += buf;
    }
#ifdef PRL_WITH_LEPTONICA
    {   // prl::backgroundNormalization = pixBackgroundNormSimple(pix, NULL, NULL) on an 8 bpp PIX
        cv::Mat gray = make_page(333, 421, 17);
        PIX* pix = pixCreate(gray.cols, gray.rows, 8);
        for (int y = 0; y < gray.rows; ++y)
            for (int x = 0; x < gray.cols; ++x) pixSetPixel(pix, x, y, gray.at<unsigned char>(y, x));
        PIX* norm = pixBackgroundNormSimple(pix, nullptr, nullptr);
        std::vector<unsigned char> want((size_t)gray.cols * gray.rows);
        prl_oracle_bgnorm(1, gray.data, gray.step, gray.cols, gray.rows, want.data(), (size_t)gray.cols);
        size_t bad = 0;
        for (int y = 0; y < gray.rows; ++y)
            for (int x = 0; x < gray.cols; ++x) {
                l_uint32 v = 0;
                pixGetPixel(norm, x, y, &v);
                bad += v != want[(size_t)y * gray.cols + x];
            }
        pixDestroy(&pix);
        pixDestroy(&norm);
        char buf[96];
        std::snprintf(buf, sizeof buf, ", \"leptonica\": \"%s\", \"bgnorm\": {\"pixels\": %d, \"mismatching\": %zu}",
                      getLeptonicaVersion(), gray.cols * gray.rows, bad);
        js += buf;
    }
#else
    js += ", \"leptonica\": null";
#endif
    js += "}";
    std::printf("%s\n", js.c_str());
    return 0;
}
#endif
