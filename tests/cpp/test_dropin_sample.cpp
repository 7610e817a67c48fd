// test_dropin_sample.cpp — a PRLib caller that keeps its own #include lines: the six per-function headers of the reference
// (src/binarizations/binarizeSauvola.h ... src/denoise/denoiseNLM.h) are included BY THEIR REFERENCE NAMES and found through
// `-I include/prl` alone.  The body is the pattern of samples/binarizations/binarizeSauvola_sample.cpp:48-53 (Mat in, one
// call with the header defaults, Mat out) with a synthetic page where the sample calls cv::imread (highgui is not part of the
// hot path, and neither box has OpenCV).
//   test_dropin_sample cpu : compiles + links against the drop-in headers; the call fails loudly without a device
//   test_dropin_sample gpu : every default call equals the CPU oracle
#include "binarizeSauvola.h"
#include "binarizeNiblack.h"
#include "binarizeWolfJolion.h"
#include "binarizeNICK.h"
#include "binarizeFeng.h"
#include "denoiseNLM.h"
#include "thinZhangSuen.h"
#include "thinGuoHall.h"
#include "backgroundNormalization.h"
#include "deskew.h"
#include "rotate.h"
#include "binarizeByLocalVariances.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../oracle/prl_oracle.h"

static cv::Mat page(int rows, int cols)
{
    cv::Mat m(rows, cols, CV_8UC1);
    unsigned s = 4711u;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            s = s * 1664525u + 1013904223u;
            int v = 205 + (int)((s >> 24) % 31) - 15;
            if (((x / 11) + (y / 6)) % 6 == 0) v -= 130;
            m.ptr(y)[x] = (unsigned char)(v < 0 ? 0 : v);
        }
    return m;
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
    int failures = 0;
    typedef void (*Fn)(cv::Mat&, cv::Mat&);
    struct Case { const char* name; int method; Fn call; } cases[] = {
        {"binarizeSauvola", PRL_SAUVOLA, [](cv::Mat& i, cv::Mat& o) { prl::binarizeSauvola(i, o); }},
        {"binarizeNiblack", PRL_NIBLACK, [](cv::Mat& i, cv::Mat& o) { prl::binarizeNiblack(i, o); }},
        {"binarizeWolfJolion", PRL_WOLFJOLION, [](cv::Mat& i, cv::Mat& o) { prl::binarizeWolfJolion(i, o); }},
        {"binarizeNICK", PRL_NICK, [](cv::Mat& i, cv::Mat& o) { prl::binarizeNICK(i, o); }},
        {"binarizeFeng", PRL_FENG, [](cv::Mat& i, cv::Mat& o) { prl::binarizeFeng(i, o); }},
    };
    for (const Case& c : cases) {
        cv::Mat inputImage = page(330, 410);
        const cv::Mat keep = inputImage.clone();
        cv::Mat outputImage;
        bool threw = false;
        std::string what;
        try {
            c.call(inputImage, outputImage);   // the sample's one line, header defaults
        } catch (const cv::Exception& e) {
            threw = true;
            what = e.what();
        }
        if (!gpu) {   // no device: the product must say so, not fall back to anything
            if (!threw || what.find("no usable HIP device") == std::string::npos) {
                std::printf("FAIL %s: expected a loud failure without a device, got '%s'\n", c.name, what.c_str());
                ++failures;
            }
            continue;
        }
        // the reference's header defaults, written out (binarizeSauvola.h:43-47, binarizeNICK.h:43-47, binarizeFeng.h:46-53)
        prl_binarize_params p{};
        p.method = c.method;
        const bool small_win = c.method == PRL_NICK || c.method == PRL_FENG;
        p.window_size = small_win ? 21 : 101;
        p.k = c.method == PRL_NICK ? -0.01 : (c.method == PRL_FENG ? 0.0 : 0.01);
        p.morph_iterations = c.method == PRL_NICK ? 0 : 2;
        p.feng_alpha1 = 0.75; p.feng_k1 = 0.2; p.feng_k2 = 0.03; p.feng_gamma = 2.0;
        prl_binarize_geometry g{};
        prl_oracle_binarize_geometry(&p, keep.cols, keep.rows, &g);
        std::vector<unsigned char> want((size_t)g.out_w * g.out_h);
        prl_oracle_binarize(&p, keep.data, keep.step, keep.cols, keep.rows, want.data(), (size_t)g.out_w);
        size_t bad = threw || outputImage.rows != g.out_h || outputImage.cols != g.out_w ? 1 : 0;
        for (int y = 0; !bad && y < outputImage.rows; ++y) bad += std::memcmp(outputImage.ptr(y), &want[(size_t)y * g.out_w], (size_t)g.out_w) != 0;
        if (bad) {
            std::printf("FAIL %s: differs from the oracle (%s)\n", c.name, what.c_str());
            ++failures;
        }
    }
    std::printf(failures ? "dropin sample: FAILED\n" : "dropin sample: OK\n");
    return failures ? 1 : 0;
}
