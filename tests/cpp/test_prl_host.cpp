// test_prl_host.cpp — exercises the C++ host layer (prlib_amd/csrc/prl/prl.h) the way a PRLib caller
// would (samples/binarizations/binarizeSauvola_sample.cpp:48-53): Mat in, prl::binarizeX(in, out, ...),
// Mat out — and checks the result, the output size and the input side effect against the CPU oracle.
//   test_prl_host cpu : argument validation, exceptions, loud failure without a device
//   test_prl_host gpu : full parity on a device
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../oracle/prl_oracle.h"
#include "../../prlib_amd/csrc/prl/prl.h"

static int g_failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            ++g_failures;                                                    \
        }                                                                    \
    } while (0)

static cv::Mat synth_page(int rows, int cols, unsigned seed, int channels = 1)
{
    cv::Mat m(rows, cols, CV_MAKETYPE(CV_8U, channels));
    unsigned s = seed * 2654435761u + 12345u;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols * channels; ++x) {
            s = s * 1664525u + 1013904223u;
            int v = 200 + (int)((s >> 24) % 41) - 20;
            if (((x / channels / 9) + (y / 5)) % 7 == 0) v -= 120;  // dark strokes
            m.ptr(y)[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    return m;
}

template <typename F> static bool throws_invalid_argument(F f)
{
    try {
        f();
    } catch (const std::invalid_argument&) {
        return true;
    } catch (...) {
        return false;
    }
    return false;
}

static void test_validation()
{
    cv::Mat empty, out;
    CHECK(throws_invalid_argument([&] { prl::binarizeSauvola(empty, out); }));
    CHECK(throws_invalid_argument([&] { prl::binarizeFeng(empty, out); }));
    cv::Mat page = synth_page(40, 50, 1);
    CHECK(throws_invalid_argument([&] { prl::binarizeSauvola(page, out, 30); }));   // even window
    CHECK(throws_invalid_argument([&] { prl::binarizeNiblack(page, out, 1); }));    // window <= 1
    CHECK(throws_invalid_argument([&] { prl::binarizeNICK(page, out, -3); }));
    CHECK(page.rows == 40 && page.cols == 50);  // a rejected call leaves the input alone
    CHECK(throws_invalid_argument([&] { prl::thinZhangSuen(empty, out); }));
    cv::Mat two(8, 8, CV_8UC2);
    CHECK(throws_invalid_argument([&] { prl::thinGuoHall(two, out); }));
    CHECK(throws_invalid_argument([&] { prl::backgroundNormalization(empty, out); }));  // backgroundNormalization.cpp:40-43
    CHECK(throws_invalid_argument([&] { prl::binarizeByLocalVariances(empty, out); }));
    CHECK(throws_invalid_argument([&] { prl::binarizeByLocalVariancesWithoutFilters(empty, out); }));
    bool threw_cv = false;
    try { prl::deskew(empty, out); } catch (const cv::Exception&) { threw_cv = true; } catch (...) {}
    CHECK(threw_cv);  // CV_Assert(!inputImage.empty()), deskew.cpp:210
    CHECK(prl::findAngle(empty) == 0.0 && prl::findOrientation(empty) == 0.0);   // no lines on an empty page, deskew.cpp:154-157
    threw_cv = false;
    cv::Mat bgr(16, 16, CV_8UC3);
    try { prl::findAngle(bgr); } catch (const cv::Exception& e) { threw_cv = e.code == cv::Error::StsAssert; } catch (...) {}
    CHECK(threw_cv);  // [upstream] cv::HoughLinesP: CV_Assert(image.type() == CV_8UC1)
    // global Otsu plumbing (host only): bimodal page splits between the modes
    cv::Mat bi(64, 64, CV_8UC1), bo;
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x) bi.at<unsigned char>(y, x) = (x < 20) ? 40 + (y % 3) : 210 + (x % 5);
    prl::binarize(bi, bo);
    CHECK(bo.rows == 64 && bo.cols == 64);
    CHECK(bo.at<unsigned char>(5, 5) == 0 && bo.at<unsigned char>(5, 40) == 255);
    std::vector<unsigned char> ob(64 * 64);
    prl_oracle_otsu(bi.data, bi.step, 64, 64, ob.data(), 64);
    CHECK(std::memcmp(ob.data(), bo.data, ob.size()) == 0);
    // BASELINE config 1 at its stated size: global Otsu on one 512 x 512 grayscale page via prl::binarize (host plumbing, no GPU),
    // on a document-like page and on a BGR page (the reference converts with cvtColor first)
    for (int ch : {1, 3}) {
        cv::Mat pg = synth_page(512, 512, 77, ch), res;
        cv::Mat gray;
        if (ch == 1) gray = pg.clone();
        else {
            gray.create(512, 512, CV_8UC1);
            prl_oracle_bgr2gray(pg.data, pg.step, 512, 512, ch, gray.data, gray.step);
        }
        std::vector<unsigned char> want((size_t)512 * 512);
        const int thr = prl_oracle_otsu(gray.data, gray.step, 512, 512, want.data(), 512);
        CHECK(thr > 60 && thr < 200);   // between the stroke and the paper mode
        prl::binarize(pg, res);
        CHECK(res.rows == 512 && res.cols == 512 && res.type() == CV_8UC1);
        size_t bad = 0, white = 0;
        for (int y = 0; y < 512 && res.cols == 512; ++y) {
            bad += std::memcmp(res.ptr(y), &want[(size_t)y * 512], 512) != 0;
            for (int x = 0; x < 512; ++x) white += res.ptr(y)[x] == 255;
        }
        CHECK(bad == 0);
        CHECK(white > 512 * 512 / 2 && white < 512 * 512);   // paper white, strokes black
    }
}

static void test_no_device_is_loud()
{
    cv::Mat page = synth_page(64, 64, 2), out;
    bool threw = false;
    try {
        prl::binarizeSauvola(page, out, 15, 0.34, 0);
    } catch (const cv::Exception& e) {
        threw = std::string(e.what()).find("no usable HIP device") != std::string::npos;
    } catch (...) {
    }
    CHECK(threw);
}

static void check_method(int method, int rows, int cols, int w, double k, int morph, int channels = 1)
{
    cv::Mat in = synth_page(rows, cols, 7u + method, channels);
    cv::Mat gray_ref;
    if (channels == 1) gray_ref = in.clone();
    else {
        gray_ref.create(rows, cols, CV_8UC1);
        prl_oracle_bgr2gray(in.data, in.step, cols, rows, channels, gray_ref.data, gray_ref.step);
    }
    prl_binarize_params p{};
    p.method = method;
    p.window_size = w;
    p.k = k;
    p.morph_iterations = morph;
    p.feng_alpha1 = 0.75;
    p.feng_k1 = 0.2;
    p.feng_k2 = 0.03;
    p.feng_gamma = 2.0;
    prl_binarize_geometry g{};
    CHECK(prl_oracle_binarize_geometry(&p, cols, rows, &g) == PRL_OK);
    std::vector<unsigned char> want((size_t)g.out_w * g.out_h), pad((size_t)g.padded_w * g.padded_h);
    CHECK(prl_oracle_binarize(&p, gray_ref.data, gray_ref.step, cols, rows, want.data(), (size_t)g.out_w) == PRL_OK);
    prl_oracle_pad_replicate(gray_ref.data, gray_ref.step, cols, rows, g.half, pad.data(), (size_t)g.padded_w);

    cv::Mat out;
    switch (method) {
    case PRL_SAUVOLA: prl::binarizeSauvola(in, out, w, k, morph); break;
    case PRL_NIBLACK: prl::binarizeNiblack(in, out, w, k, morph); break;
    case PRL_WOLFJOLION: prl::binarizeWolfJolion(in, out, w, k, morph); break;
    case PRL_NICK: prl::binarizeNICK(in, out, w, k, morph); break;
    default: prl::binarizeFeng(in, out, w, 0.75, 0.2, 0.03, 2.0, morph); break;
    }
    CHECK(out.rows == g.out_h && out.cols == g.out_w && out.type() == CV_8UC1);
    size_t bad = 0;
    for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * g.out_w], (size_t)g.out_w) != 0;
    CHECK(bad == 0);
    // side effect: the caller's Mat is now the replicate-padded gray page (binarizeSauvola.cpp:51,65)
    CHECK(in.rows == g.padded_h && in.cols == g.padded_w && in.channels() == 1);
    bad = 0;
    for (int y = 0; y < in.rows && in.cols == g.padded_w; ++y)
        bad += std::memcmp(in.ptr(y), &pad[(size_t)y * g.padded_w], (size_t)g.padded_w) != 0;
    CHECK(bad == 0);
}

static void test_gpu()
{
    check_method(PRL_SAUVOLA, 120, 160, 15, 0.34, 0);
    check_method(PRL_SAUVOLA, 300, 260, 101, 0.01, 2);  // header defaults
    check_method(PRL_NIBLACK, 97, 131, 31, 0.2, 2);
    check_method(PRL_WOLFJOLION, 140, 150, 31, 0.3, -1);
    check_method(PRL_NICK, 111, 99, 21, -0.01, 0);
    check_method(PRL_FENG, 130, 128, 21, 0.0, 2);
    check_method(PRL_SAUVOLA, 90, 100, 15, 0.34, 0, 3);  // BGR input: cvtColor in place, then padded
    // ROI input (step > cols), as cv::Mat views are
    cv::Mat big = synth_page(100, 200, 9), roi = big(cv::Rect(10, 5, 120, 80)), out;
    cv::Mat roi_copy = roi.clone();
    prl::binarizeNICK(roi, out, 21, -0.1, 0);
    prl_binarize_params p{};
    p.method = PRL_NICK; p.window_size = 21; p.k = -0.1;
    std::vector<unsigned char> want((size_t)(120 - 21) * (80 - 21));
    CHECK(prl_oracle_binarize(&p, roi_copy.data, roi_copy.step, 120, 80, want.data(), 120 - 21) == PRL_OK);
    CHECK(out.rows == 59 && out.cols == 99);
    size_t bad = 0;
    for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * 99], 99) != 0;
    CHECK(bad == 0);
    // Wolf/NICK/Feng on a page no larger than the window: cv::Exception upstream (empty ROI)
    cv::Mat small = synth_page(21, 40, 3);
    bool threw = false;
    try { prl::binarizeNICK(small, out, 21); } catch (const cv::Exception&) { threw = true; } catch (...) {}
    CHECK(threw);
    // prl::denoise on a BGR scan
    cv::Mat noisy = synth_page(70, 90, 11, 3), den;
    prl::denoise(noisy, den, 10.0);
    std::vector<unsigned char> dw((size_t)70 * 90 * 3);
    CHECK(prl_oracle_denoise(3, 10.0f, noisy.data, noisy.step, 90, 70, dw.data(), 270, 4) == PRL_OK);
    CHECK(den.rows == 70 && den.cols == 90 && den.channels() == 3);
    bad = 0;
    for (int y = 0; y < 70; ++y) bad += std::memcmp(den.ptr(y), &dw[(size_t)y * 270], 270) != 0;
    CHECK(bad == 0);
    threw = false;
    cv::Mat gray1 = synth_page(30, 30, 5);
    try { prl::denoise(gray1, den); } catch (const cv::Exception&) { threw = true; } catch (...) {}
    CHECK(threw);  // 8UC1 is rejected by fastNlMeansDenoisingColored
    // thinning: binarize, invert (white = foreground for prl::thin*), thin
    cv::Mat pg = synth_page(140, 180, 13), mk, sk;
    prl::binarizeSauvola(pg, mk, 15, 0.34, 0);
    for (int y = 0; y < mk.rows; ++y)
        for (int x = 0; x < mk.cols; ++x) mk.at<unsigned char>(y, x) = 255 - mk.at<unsigned char>(y, x);
    std::vector<unsigned char> tw((size_t)mk.rows * mk.cols);
    for (int method = 0; method < 2; ++method) {
        CHECK(prl_oracle_thin(method, mk.data, mk.step, mk.cols, mk.rows, tw.data(), (size_t)mk.cols, nullptr) == PRL_OK);
        if (method == 0) prl::thinZhangSuen(mk, sk); else prl::thinGuoHall(mk, sk);
        CHECK(sk.rows == mk.rows && sk.cols == mk.cols);
        bad = 0;
        for (int y = 0; y < sk.rows; ++y) bad += std::memcmp(sk.ptr(y), &tw[(size_t)y * mk.cols], (size_t)mk.cols) != 0;
        CHECK(bad == 0);
    }
    // in place (input and output share their data, thinZhangSuen.cpp:71-74): the skeleton is written THROUGH the caller's buffer
    // (:100-103) - same data pointer, same step afterwards, also for a ROI view; a 3-channel in-place call changes nothing (the
    // reference's cvtColor re-allocates its working Mat, :84, and the result is dropped)
    {
        CHECK(prl_oracle_thin(0, mk.data, mk.step, mk.cols, mk.rows, tw.data(), (size_t)mk.cols, nullptr) == PRL_OK);
        cv::Mat wide(mk.rows, mk.cols + 24, CV_8UC1);
        std::memset(wide.data, 7, (size_t)wide.rows * wide.step);
        cv::Mat view = wide(cv::Rect(11, 0, mk.cols, mk.rows));
        for (int y = 0; y < mk.rows; ++y) std::memcpy(view.ptr(y), mk.ptr(y), (size_t)mk.cols);
        cv::Mat same = view;
        unsigned char* before = view.data;
        prl::thinZhangSuen(view, same);
        CHECK(view.data == before && same.data == before && view.step == wide.step && view.cols == mk.cols);
        bad = 0;
        for (int y = 0; y < mk.rows; ++y) {
            bad += std::memcmp(view.ptr(y), &tw[(size_t)y * mk.cols], (size_t)mk.cols) != 0;
            bad += wide.at<unsigned char>(y, 10) != 7 || wide.at<unsigned char>(y, 11 + mk.cols) != 7;   // nothing outside the view
        }
        CHECK(bad == 0);
        cv::Mat c3(20, 30, CV_8UC3);
        std::memset(c3.data, 200, (size_t)c3.rows * c3.step);
        cv::Mat c3same = c3;
        prl::thinGuoHall(c3, c3same);
        CHECK(c3.type() == CV_8UC3 && c3same.data == c3.data && c3.at<unsigned char>(5, 7) == 200);
    }
    // prl::findOrientation (deskew.h:52): 0.0 for the 1-channel page prl::deskew hands it (deskew.cpp:238, :73-84); a 3-channel
    // caller would get Leptonica's detector in the reference - here a cv::Exception that says so, not a silent "up"
    {
        cv::Mat g1 = synth_page(40, 50, 3);
        CHECK(prl::findOrientation(g1) == 0.0);
        cv::Mat c3(40, 50, CV_8UC3);
        std::memset(c3.data, 128, (size_t)c3.rows * c3.step);
        bool not_impl = false;
        try { (void)prl::findOrientation(c3); } catch (const cv::Exception& e) { not_impl = e.code == cv::Error::StsNotImplemented && std::strstr(e.what(), "pixOrientDetectDwa") != nullptr; }
        CHECK(not_impl);
    }
}

// a page of horizontal "text lines" drawn at a small slope, for deskew
static cv::Mat text_page(int rows, int cols, double slope, int channels)
{
    cv::Mat m(rows, cols, CV_MAKETYPE(CV_8U, channels));
    unsigned s = 99u;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            s = s * 1664525u + 1013904223u;
            const double v = y - slope * x;
            const int line = (int)(v / 30.0);
            const bool ink = v > 20 && v < rows - 20 && x > 15 && x < cols - 15 && (v - 30.0 * line) < 8.0 && ((x + 17 * line) % 60) < 45;
            const int val = ink ? 40 + (int)((s >> 24) % 20) : 215 + (int)((s >> 24) % 20);
            for (int c = 0; c < channels; ++c) m.ptr(y)[x * channels + c] = (unsigned char)(val - 5 * c);
        }
    return m;
}

static void test_gpu_round2()
{
    // prl::backgroundNormalization: gray and BGR
    for (int ch : {1, 3, 4}) {
        cv::Mat in = synth_page(155, 212, 21, ch), out;
        const int och = ch == 1 ? 1 : 3;
        std::vector<unsigned char> want((size_t)155 * 212 * och);
        CHECK(prl_oracle_bgnorm(ch, in.data, in.step, 212, 155, want.data(), (size_t)212 * och) == PRL_OK);
        prl::backgroundNormalization(in, out);
        CHECK(out.rows == 155 && out.cols == 212 && out.channels() == och);
        size_t bad = 0;
        for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * 212 * och], (size_t)212 * och) != 0;
        CHECK(bad == 0);
    }
    // prl::rotate: general angle (square canvas) and a quarter turn (transposed)
    {
        cv::Mat in = synth_page(80, 130, 5, 3), out;
        std::vector<unsigned char> want((size_t)130 * 130 * 3);
        CHECK(prl_oracle_rotate(3, in.data, in.step, 130, 80, 7.5, want.data(), 130 * 3) == PRL_OK);
        prl::rotate(in, out, 7.5);
        CHECK(out.rows == 130 && out.cols == 130 && out.channels() == 3);
        size_t bad = 0;
        for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * 390], 390) != 0;
        CHECK(bad == 0);
        prl::rotate(in, out, 90.0);
        CHECK(out.rows == 130 && out.cols == 80);
        CHECK(prl_oracle_rotate(3, in.data, in.step, 130, 80, 90.0, want.data(), 80 * 3) == PRL_OK);
        bad = 0;
        for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * 240], 240) != 0;
        CHECK(bad == 0);
    }
    // prl::deskew: a skewed text page comes back square, identical to the oracle; a blank page comes back as a clone
    for (int ch : {1, 3}) {
        cv::Mat in = text_page(300, 420, 0.04, ch), out;
        std::vector<unsigned char> want((size_t)420 * 420 * ch);
        int ow = 0, oh = 0, nl = 0;
        double ang = 0;
        CHECK(prl_oracle_deskew(ch, in.data, in.step, 420, 300, want.data(), (size_t)420 * ch, &ow, &oh, &ang, nullptr, &nl) == PRL_OK);
        CHECK(nl > 5 && ang != 0.0 && ow == 420 && oh == 420);
        CHECK(prl::deskew(in, out));
        CHECK(out.rows == oh && out.cols == ow && out.channels() == ch);
        size_t bad = 0;
        for (int y = 0; y < out.rows && out.cols == ow; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * ow * ch], (size_t)ow * ch) != 0;
        CHECK(bad == 0);
    }
    // prl::findAngle (deskew.h:62) on the thresholded page prl::deskew would hand it (deskew.cpp:224-226): the oracle's angle,
    // through a continuous Mat and through a ROI view (step > cols); prl::findOrientation (deskew.h:52) is 0.0
    {
        cv::Mat gray = text_page(300, 420, 0.04, 1), bin(300, 420, CV_8UC1);
        prl_oracle_otsu(gray.data, gray.step, 420, 300, bin.data, bin.step);
        int nl = 0;
        const double want = prl_oracle_find_angle(bin.data, bin.step, 420, 300, &nl);
        CHECK(nl > 5 && want != 0.0);
        const cv::Mat keep = bin.clone();
        CHECK(prl::findAngle(bin) == want);
        CHECK(std::memcmp(bin.data, keep.data, (size_t)300 * 420) == 0);   // const input: untouched (the reference clones, :144)
        cv::Mat wide(300, 500, CV_8UC1);
        std::memset(wide.data, 255, (size_t)300 * 500);
        cv::Mat view = wide(cv::Rect(37, 0, 420, 300));
        bin.copyTo(view);
        CHECK(view.data == wide.data + 37 && !view.isContinuous());
        CHECK(prl::findAngle(view) == want);
        cv::Mat white(200, 300, CV_8UC1);
        std::memset(white.data, 255, (size_t)200 * 300);
        CHECK(prl::findAngle(white) == 0.0);   // no points, no segment: 0.0 (:154-157)
        CHECK(prl::findOrientation(bin) == 0.0);
    }
    // prl::binarizeByLocalVariancesWithoutFilters (integer-exact) and binarizeByLocalVariances (float32 log / exp: tolerance)
    {
        cv::Mat in = synth_page(140, 190, 31, 3), out;
        std::vector<unsigned char> want((size_t)140 * 190);
        CHECK(prl_oracle_binarize_lv_nofilters(in.data, in.step, 190, 140, 0.125, 10, want.data(), 190) == PRL_OK);
        prl::binarizeByLocalVariancesWithoutFilters(in, out);
        CHECK(out.rows == 140 && out.cols == 190 && out.type() == CV_8UC1);
        size_t bad = 0;
        for (int y = 0; y < out.rows; ++y) bad += std::memcmp(out.ptr(y), &want[(size_t)y * 190], 190) != 0;
        CHECK(bad == 0);
        CHECK(prl_oracle_binarize_lv(in.data, in.step, 190, 140, 0.125, 25, 2.0, want.data(), 190) == PRL_OK);
        prl::binarizeByLocalVariances(in, out);
        size_t diff = 0;
        for (int y = 0; y < out.rows; ++y)
            for (int x = 0; x < out.cols; ++x) diff += out.ptr(y)[x] != want[(size_t)y * 190 + x];
        CHECK(diff <= 27);  // 1e-3 of the page
        cv::Mat gray1 = synth_page(20, 20, 3, 1);
        CHECK(throws_invalid_argument([&] { prl::binarizeByLocalVariances(gray1, out); }));
    }
    // BASELINE config 5 the way a PRLib user writes it (one prl:: call per stage and page, through host Mats) against the
    // one-call chain on the same host pages (prl_hip_chain_batch_host): identical bytes, sizes and angles
    {
        const int n = 3, W = 260, H = 180;
        std::vector<cv::Mat> scans;
        scans.push_back(text_page(H, W, 0.035, 3));
        scans.push_back(text_page(H, W, 0.0, 3));
        scans.push_back(text_page(H, W, -0.05, 3));
        std::vector<cv::Mat> want(n);
        for (int i = 0; i < n; ++i) {
            cv::Mat a, b, c, d, e;
            prl::deskew(scans[i], a);
            prl::denoise(a, b, 10);
            prl::backgroundNormalization(b, c);
            prl::binarizeSauvola(c, d, 31, 0.34, 0);
            e.create(d.rows, d.cols, CV_8UC1);
            for (int y = 0; y < d.rows; ++y)
                for (int x = 0; x < d.cols; ++x) e.ptr(y)[x] = (unsigned char)(255 - d.ptr(y)[x]);   // cv::bitwise_not
            prl::thinZhangSuen(e, want[i]);
        }
        prl_chain_params cp;
        prl_hip_default_chain_params(&cp);
        cp.deskew = 1;
        cp.denoise = 1; cp.denoise_strength = 10.f;
        cp.background_normalization = 1;
        cp.binarize.window_size = 31; cp.binarize.k = 0.34; cp.binarize.morph_iterations = 0;
        cp.thin = PRL_THIN_ZHANGSUEN;
        int mw = 0, mh = 0;
        CHECK(prl_hip_chain_max_out_size(&cp, W, H, &mw, &mh) == PRL_OK);
        std::vector<cv::Mat> out(n);
        std::vector<const uint8_t*> src(n);
        std::vector<uint8_t*> dst(n);
        for (int i = 0; i < n; ++i) { src[i] = scans[i].data; out[i].create(mh, mw, CV_8UC1); dst[i] = out[i].data; }
        std::vector<int32_t> wh(2 * n);
        std::vector<double> angle(n);
        CHECK(prl_hip_chain_batch_host(&cp, n, 3, src.data(), scans[0].step, W, H, dst.data(), out[0].step, wh.data(), angle.data(), 0) == PRL_OK);
        int rotated = 0;
        for (int i = 0; i < n; ++i) {
            CHECK(wh[2 * i] == want[i].cols && wh[2 * i + 1] == want[i].rows);
            size_t bad = 0;
            for (int y = 0; y < want[i].rows && wh[2 * i] == want[i].cols; ++y) bad += std::memcmp(out[i].ptr(y), want[i].ptr(y), (size_t)want[i].cols) != 0;
            CHECK(bad == 0);
            rotated += angle[i] != 0.0;
        }
        CHECK(rotated >= 2);
    }
    cv::Mat blank(60, 90, CV_8UC1), bout;
    std::memset(blank.data, 230, 60 * 90);
    CHECK(prl::deskew(blank, bout) && bout.rows == 60 && bout.cols == 90 && std::memcmp(bout.data, blank.data, 60 * 90) == 0);
}

int main(int argc, char** argv)
{
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    test_validation();
    if (mode == "gpu") { test_gpu(); test_gpu_round2(); }
    else test_no_device_is_loud();
    if (g_failures) {
        std::printf("%d check(s) failed\n", g_failures);
        return 1;
    }
    std::printf("ok (%s)\n", mode.c_str());
    return 0;
}
