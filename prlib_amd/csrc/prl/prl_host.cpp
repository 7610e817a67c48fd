// prl_host.cpp — thin C++ shims: validate like the reference, hand (data, step, rows, cols) to the C ABI.
//
// Builds against real OpenCV (PRL_HAVE_OPENCV, set by prl.h when <opencv2/core/core.hpp> exists) and against cvmat_shim.h.
// Only names both have are used; tests/test_cpp_host.py compiles this file against OpenCV's declared signatures
// (tests/cpp/opencv_api/).  With OpenCV present the colour conversion is OpenCV's own cv::cvtColor, as in the reference.
#include "prl.h"

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef PRL_HAVE_OPENCV
#include <opencv2/imgproc/imgproc.hpp>   // cv::cvtColor — the header the reference includes (binarizeSauvola.cpp:29)
#endif

#include "../../../include/prl_hip.h"

namespace {

// Every cv::Exception of this layer is thrown here, through the constructor OpenCV has:
// Exception(int code, const String& err, const String& func, const String& file, int line).
[[noreturn]] void fail_cv(int code, const std::string& err, const char* func, int line)
{
    throw cv::Exception(code, err, func, __FILE__, line);
}
#define PRL_FAIL_CV(code, err) fail_cv((code), (err), __func__, __LINE__)

// A C-ABI status as the exception the reference would raise at that point.
[[noreturn]] void raise_at(int status, const char* func, int line)
{
    const std::string msg = prl_hip_strerror(status);
    if (status == PRL_ERR_EMPTY || status == PRL_ERR_BAD_WINDOW)
        throw std::invalid_argument(msg);  // binarizeSauvola.cpp:38-47
    const std::string detail = prl_hip_last_error_detail();
    int code = cv::Error::StsError;
    switch (status) {
    case PRL_ERR_BAD_CHANNELS: code = cv::Error::StsUnsupportedFormat; break;   // cvtColor / NLM reject the type
    case PRL_ERR_EMPTY_RECT: code = cv::Error::StsAssert; break;                // the ROI assertion of cv::Mat::operator()
    case PRL_ERR_BAD_ARG: code = cv::Error::StsBadArg; break;
    case PRL_ERR_NO_DEVICE:
    case PRL_ERR_HIP: code = cv::Error::GpuApiCallError; break;
    case PRL_ERR_NOMEM: code = cv::Error::StsNoMem; break;
    default: break;
    }
    fail_cv(code, detail.empty() ? msg : msg + " [" + detail + "]", func, line);
}
#define raise(status) raise_at((status), __func__, __LINE__)

// cv::cvtColor(in, in, cv::COLOR_BGR2GRAY) on 8-bit BGR/BGRA — binarizeSauvola.cpp:51.
// With OpenCV: the installed cv::cvtColor itself (its luma coefficients differ between versions, SURVEY.md Appendix B, and
// the reference gets whatever the machine has).  Without: the 14-bit fixed-point luma, a per-pixel conversion on the host.
void bgr2gray_inplace(cv::Mat& m)
{
#ifdef PRL_HAVE_OPENCV
    cv::cvtColor(m, m, cv::COLOR_BGR2GRAY);
#else
    const int cn = m.channels();
    cv::Mat g(m.rows, m.cols, CV_8UC1);
    for (int y = 0; y < m.rows; ++y) {
        const unsigned char* s = m.ptr(y);
        unsigned char* d = g.ptr(y);
        for (int x = 0; x < m.cols; ++x, s += cn)
            d[x] = (unsigned char)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
    }
    m = g;
#endif
}

void run(int method, cv::Mat& in, cv::Mat& out, int windowSize, double k, int morph, double a1 = 0.75,
         double k1 = 0.2, double k2 = 0.03, double gamma = 2.0)
{
    if (in.empty()) throw std::invalid_argument("Input image for binarization is empty");
    if (!((windowSize > 1) && ((windowSize % 2) == 1)))
        throw std::invalid_argument(prl_hip_strerror(PRL_ERR_BAD_WINDOW));
    if (in.depth() != CV_8U) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "prl: 8-bit images only");
    if (in.channels() != 1) {
        if (in.channels() != 3 && in.channels() != 4) raise(PRL_ERR_BAD_CHANNELS);
        bgr2gray_inplace(in);
    }
    prl_binarize_params p{};
    p.method = method;
    p.window_size = windowSize;
    p.k = k;
    p.morph_iterations = morph;
    p.feng_alpha1 = a1;
    p.feng_k1 = k1;
    p.feng_k2 = k2;
    p.feng_gamma = gamma;
    prl_binarize_geometry g{};
    int st = prl_hip_binarize_geometry(&p, in.cols, in.rows, &g);
    if (st != PRL_OK) raise(st);
    cv::Mat result(g.out_h, g.out_w, CV_8UC1);
#ifndef PRL_KEEP_INPUT
    cv::Mat padded(g.padded_h, g.padded_w, CV_8UC1);
    st = prl_hip_binarize_host(&p, in.data, in.step, in.cols, in.rows, result.data, result.step, padded.data,
                               padded.step);
#else
    st = prl_hip_binarize_host(&p, in.data, in.step, in.cols, in.rows, result.data, result.step, nullptr, 0);
#endif
    if (st != PRL_OK) raise(st);
#ifndef PRL_KEEP_INPUT
    in = padded;  // cv::copyMakeBorder(in, in, ...) — binarizeSauvola.cpp:65
#endif
    out = result;  // outputImage = in(rect) > T — :122
}

}  // namespace

void prl::binarizeSauvola(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize,
                          double thresholdCoefficient, int morphIterationCount)
{
    run(PRL_SAUVOLA, inputImage, outputImage, windowSize, thresholdCoefficient, morphIterationCount);
}

void prl::binarizeNiblack(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize,
                          double thresholdCoefficient, int morphIterationCount)
{
    run(PRL_NIBLACK, inputImage, outputImage, windowSize, thresholdCoefficient, morphIterationCount);
}

void prl::binarizeWolfJolion(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize,
                             double thresholdCoefficient, int morphIterationCount)
{
    run(PRL_WOLFJOLION, inputImage, outputImage, windowSize, thresholdCoefficient, morphIterationCount);
}

void prl::binarizeNICK(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize, double thresholdCoefficient,
                       int morphIterationCount)
{
    run(PRL_NICK, inputImage, outputImage, windowSize, thresholdCoefficient, morphIterationCount);
}

void prl::binarizeFeng(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize,
                       double thresholdCoefficient_alpha1, double thresholdCoefficient_k1,
                       double thresholdCoefficient_k2, double thresholdCoefficient_gamma, int morphIterationCount)
{
    run(PRL_FENG, inputImage, outputImage, windowSize, 0.0, morphIterationCount, thresholdCoefficient_alpha1,
        thresholdCoefficient_k1, thresholdCoefficient_k2, thresholdCoefficient_gamma);
}

void prl::denoise(const cv::Mat& inputImage, cv::Mat& outputImage, double strength)
{
    // [upstream] an empty Mat has type CV_8UC1: "Type of input image should be CV_8UC3 or CV_8UC4!"
    if (inputImage.empty()) raise(PRL_ERR_BAD_CHANNELS);
    if (inputImage.depth() != CV_8U) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "prl::denoise: 8-bit images only");
    cv::Mat result(inputImage.rows, inputImage.cols, inputImage.type());
    const int st = prl_hip_denoise_host(inputImage.channels(), (float)strength, inputImage.data, inputImage.step,
                                        inputImage.cols, inputImage.rows, result.data, result.step);
    if (st != PRL_OK) raise(st);  // 1/2-channel input: "Type of input image should be CV_8UC3 or CV_8UC4!" upstream
    outputImage = result;
}

namespace {
void lv_impl(int with_filters, cv::Mat& in, cv::Mat& out, double coeff, int minVar, double gamma, const char* empty_msg)
{
    if (in.empty()) throw std::invalid_argument(empty_msg);
    // [upstream] MatToLocalVarianceMap accepts 8UC1 / 8UC3 (imageLibCommon.cpp:404-408), but the callers index three planes
    if (in.type() != CV_8UC3)
        throw std::invalid_argument("Image for local variance map extraction has unsupported type (3 channels, 8 bits required here)");
    cv::Mat result(in.rows, in.cols, CV_8UC1);
    const int st = prl_hip_binarize_lv_host(with_filters, coeff, minVar, gamma, in.data, in.step, in.cols, in.rows, result.data, result.step);
    if (st != PRL_OK) raise(st);
    out = result;
}
}  // namespace

void prl::binarizeByLocalVariances(cv::Mat& inputImage, cv::Mat& outputImage, double varianceThresholdCoeff, int minResultVariance,
                                   double gamma)
{
    lv_impl(1, inputImage, outputImage, varianceThresholdCoeff, minResultVariance, gamma,
            "binarizeByLocalVariances: Input inputImage for binarization is empty");
}

void prl::binarizeByLocalVariancesWithoutFilters(cv::Mat& inputImage, cv::Mat& outputImage, double varianceThresholdCoeff,
                                                 int minResultVariance)
{
    lv_impl(0, inputImage, outputImage, varianceThresholdCoeff, minResultVariance, 2.0,
            "binarizeByLocalVariancesWithoutFilters: Input inputImage for binarization is empty");
}

void prl::backgroundNormalization(const cv::Mat& inputImage, cv::Mat& outputImage)
{
    if (inputImage.empty()) throw std::invalid_argument("Input image for flipping is empty");  // backgroundNormalization.cpp:40-43
    if (inputImage.depth() != CV_8U) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "Cannot convert RAW image to Pix\n");   // formatConvert.cpp:103-104
    const int cn = inputImage.channels();
    if (cn != 1 && cn != 3 && cn != 4) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "Cannot convert RAW image to Pix\n");
    cv::Mat result(inputImage.rows, inputImage.cols, cn == 1 ? CV_8UC1 : CV_8UC3);
    const int st = prl_hip_bgnorm_host(cn, inputImage.data, inputImage.step, inputImage.cols, inputImage.rows, result.data,
                                       result.step);
    if (st != PRL_OK) raise(st);
    outputImage = result;
}

void prl::rotate(const cv::Mat& inputImage, cv::Mat& outputImage, double angle)
{
    if (inputImage.empty()) raise(PRL_ERR_EMPTY);  // [upstream] cv::transpose / cv::warpAffine assert on an empty source
    if (inputImage.depth() != CV_8U) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "prl::rotate: 8-bit images only");
    int ow = 0, oh = 0;
    int st = prl_hip_rotate_out_size(inputImage.cols, inputImage.rows, angle, &ow, &oh);
    if (st != PRL_OK) raise(st);
    cv::Mat result(oh, ow, inputImage.type());
    st = prl_hip_rotate_host(inputImage.channels(), angle, inputImage.data, inputImage.step, inputImage.cols, inputImage.rows,
                             result.data, result.step);
    if (st != PRL_OK) raise(st);
    outputImage = result;
}

bool prl::deskew(const cv::Mat& inputImage, cv::Mat& outputImage)
{
    if (inputImage.empty()) PRL_FAIL_CV(cv::Error::StsAssert, "!inputImage.empty()");  // CV_Assert, deskew.cpp:210
    if (inputImage.depth() != CV_8U) PRL_FAIL_CV(cv::Error::StsUnsupportedFormat, "prl::deskew: 8-bit images only");
    const int len = inputImage.cols > inputImage.rows ? inputImage.cols : inputImage.rows;
    cv::Mat big(len, len, inputImage.type());
    int ow = 0, oh = 0;
    const int st = prl_hip_deskew_host(inputImage.channels(), inputImage.data, inputImage.step, inputImage.cols, inputImage.rows,
                                       big.data, big.step, &ow, &oh, nullptr);
    if (st != PRL_OK) raise(st);
    outputImage = (ow == len && oh == len) ? big : big(cv::Rect(0, 0, ow, oh)).clone();
    return !outputImage.empty();  // deskew.cpp:245-250
}

// deskew.h:62, deskew.cpp:139-205.  The reference hands cv::HoughLinesP the complement of the page; HoughLinesP takes
// CV_8UC1 only ([upstream] CV_Assert(image.type() == CV_8UC1)) and finds no line on an empty image, so 0.0 (:154-157).
double prl::findAngle(const cv::Mat& inputImage)
{
    if (inputImage.empty()) return 0.0;
    if (inputImage.type() != CV_8UC1) PRL_FAIL_CV(cv::Error::StsAssert, "image.type() == CV_8UC1");
    double angle = 0.0;
    const int st = prl_hip_find_angle_host(inputImage.data, inputImage.step, inputImage.cols, inputImage.rows, &angle, nullptr);
    if (st != PRL_OK) raise(st);
    return angle;
}

// deskew.h:52, deskew.cpp:70-136.  The reference fills its gray page only for 3-channel input (:73-76), thresholds it
// (cv::adaptiveThreshold, :78) and asks Leptonica's pixOrientDetectDwa / makeOrientDecision for one of four orientations
// (:91, :101).  prl::deskew - its only caller (:238) - always hands it the 1-channel thresholded page, for which the gray page
// is EMPTY: the function leaves through its NULL-pix exit with 0 (:80-84) or throws inside adaptiveThreshold, depending on the
// OpenCV version.  This layer answers that 0.0 for every input that is not 3-channel: the orientation step of the hot path is a
// no-op.  For a 3-CHANNEL caller the reference really looks (and can answer 90 / 180 / 270): Leptonica's
// DWA text-orientation detector is outside this library's scope and its source is not available to restate - such
// a call fails loudly instead of answering "up" without having looked.
double prl::findOrientation(const cv::Mat& inputImage)
{
    if (!inputImage.empty() && inputImage.channels() == 3)
        PRL_FAIL_CV(cv::Error::StsNotImplemented,
                    "prl::findOrientation on a 3-channel image needs Leptonica's pixOrientDetectDwa (deskew.cpp:91), which this library does not provide");
    return 0.0;
}

namespace {
void thin_impl(int method, cv::Mat& inputImage, cv::Mat& outputImage)
{
    if (inputImage.empty()) throw std::invalid_argument("Input image for thinning is empty");
    if (inputImage.type() != CV_8UC3 && inputImage.type() != CV_8UC1)
        throw std::invalid_argument("Invalid type of image for thinning (required 8 or 24 bits per pixel)");
    // The reference works on the caller's buffer when input and output share data, else on a clone (thinZhangSuen.cpp:71-80).
    // In place, 1 channel: every step (&= 1, the iterations, *= 255 at :100-103) writes through the shared buffer; the two headers
    // stay as they are.  In place, 3 channels: cvtColor (:84) gives the working Mat a new 1-channel buffer, the skeleton is
    // computed there and dropped - neither of the caller's Mats changes (reproduced: nothing to compute).
    const bool in_place = inputImage.data == outputImage.data;
    if (in_place && inputImage.channels() == 3) return;
    cv::Mat work = in_place ? inputImage : inputImage.clone();
    if (work.channels() == 3) bgr2gray_inplace(work);
    cv::Mat result(work.rows, work.cols, CV_8UC1);
    const int st = prl_hip_thin_host(method, work.data, work.step, work.cols, work.rows, result.data, result.step);
    if (st != PRL_OK) raise(st);
    if (in_place) {
        for (int y = 0; y < result.rows; ++y) std::memcpy(inputImage.ptr(y), result.ptr(y), (size_t)result.cols);
        return;
    }
    outputImage = result;
}
}  // namespace

void prl::thinZhangSuen(cv::Mat& inputImage, cv::Mat& outputImage) { thin_impl(PRL_THIN_ZHANGSUEN, inputImage, outputImage); }
void prl::thinGuoHall(cv::Mat& inputImage, cv::Mat& outputImage) { thin_impl(PRL_THIN_GUOHALL, inputImage, outputImage); }

// Global Otsu on the host (BASELINE config 1: plumbing, no GPU).  [upstream getThreshVal_Otsu_8u]
void prl::binarize(cv::Mat& inputImage, cv::Mat& outputImage)
{
    if (inputImage.empty()) throw std::invalid_argument("Input image for binarization is empty");
    if (inputImage.channels() != 1) bgr2gray_inplace(inputImage);
    const cv::Mat& in = inputImage;
    long hist[256] = {0};
    for (int y = 0; y < in.rows; ++y) {
        const unsigned char* s = in.ptr(y);
        for (int x = 0; x < in.cols; ++x) hist[s[x]]++;
    }
    const double scale = 1.0 / ((double)in.rows * in.cols);
    double mu = 0;
    for (int i = 0; i < 256; ++i) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, best_sigma = 0;
    int best = 0;
    for (int i = 0; i < 256; ++i) {
        const double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        const double lo = q1 < q2 ? q1 : q2, hi = q1 < q2 ? q2 : q1;
        if (lo < 1.1920928955078125e-07 || hi > 1.0 - 1.1920928955078125e-07) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > best_sigma) {
            best_sigma = sigma;
            best = i;
        }
    }
    cv::Mat result(in.rows, in.cols, CV_8UC1);
    for (int y = 0; y < in.rows; ++y) {
        const unsigned char* s = in.ptr(y);
        unsigned char* d = result.ptr(y);
        for (int x = 0; x < in.cols; ++x) d[x] = s[x] > best ? 255 : 0;
    }
    outputImage = result;
}
