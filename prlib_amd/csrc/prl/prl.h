// prl.h — C++ host layer with the reference's signatures for the hot path, over the C ABI of
// include/prl_hip.h.  A caller of PRLib switches by including this header instead of the six reference
// headers and linking libprlib_hip.so + prl_host.cpp:
//
//   reference header (PRLib tree)                          function
//   src/binarizations/binarizeSauvola.h:43-47              prl::binarizeSauvola
//   src/binarizations/binarizeNiblack.h:43-47              prl::binarizeNiblack
//   src/binarizations/binarizeWolfJolion.h:43-47           prl::binarizeWolfJolion
//   src/binarizations/binarizeNICK.h:43-47                 prl::binarizeNICK
//   src/binarizations/binarizeFeng.h:46-53                 prl::binarizeFeng
//   src/denoise/denoiseNLM.h:32                            prl::denoise
//
// Same names, argument order, defaults, exceptions (std::invalid_argument for an empty image or a bad
// window, binarizeSauvola.cpp:38-47) and side effects: the caller's input Mat is converted to gray
// (:51) and replaced by the replicate-padded page (:65); the output Mat is (re)allocated (:122).
// Define PRL_KEEP_INPUT before including to opt out of the input mutation.
// include/prl/ holds one forwarding header per reference header (binarizeSauvola.h, ..., denoiseNLM.h), so a caller keeps
// its #include lines and only changes the include path.
#pragma once

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define PRL_HAVE_OPENCV 1
#endif
#endif
#if defined(PRL_REQUIRE_OPENCV) && !defined(PRL_HAVE_OPENCV)
#error "PRL_REQUIRE_OPENCV: <opencv2/core/core.hpp> was not found on the include path"
#endif
#ifndef PRL_HAVE_OPENCV
#include "cvmat_shim.h"
#endif
#ifndef CV_EXPORTS   // (OpenCV defines it; the shim build exports the same way)
#define CV_EXPORTS __attribute__((visibility("default")))
#endif

namespace prl {

CV_EXPORTS void binarizeSauvola(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize = 101,
                     double thresholdCoefficient = 0.01, int morphIterationCount = 2);

CV_EXPORTS void binarizeNiblack(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize = 101,
                     double thresholdCoefficient = 0.01, int morphIterationCount = 2);

CV_EXPORTS void binarizeWolfJolion(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize = 101,
                        double thresholdCoefficient = 0.01, int morphIterationCount = 2);

CV_EXPORTS void binarizeNICK(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize = 21,
                  double thresholdCoefficient = -0.01, int morphIterationCount = 0);

CV_EXPORTS void binarizeFeng(cv::Mat& inputImage, cv::Mat& outputImage, int windowSize = 21,
                  double thresholdCoefficient_alpha1 = 0.75, double thresholdCoefficient_k1 = 0.2,
                  double thresholdCoefficient_k2 = 0.03, double thresholdCoefficient_gamma = 2.0,
                  int morphIterationCount = 2);

CV_EXPORTS void denoise(const cv::Mat& inputImage, cv::Mat& outputImage, double strength = 5.5);

// SURVEY.md §8f rank 1 — src/thinning/thinZhangSuen.h, src/thinning/thinGuoHall.h.  8UC1 or 8UC3 (BGR is
// converted to gray first, thinZhangSuen.cpp:78-81); foreground = pixels with bit 0 set; output 0/255.
// std::invalid_argument for an empty image or another type (:59-68).
CV_EXPORTS void thinZhangSuen(cv::Mat& inputImage, cv::Mat& outputImage);
CV_EXPORTS void thinGuoHall(cv::Mat& inputImage, cv::Mat& outputImage);

// SURVEY.md §8f rank 4b — src/binarizations/binarizeByLocalVariances.h:8-12.  8UC3 input (the reference reads three
// variance planes); std::invalid_argument for an empty image (binarizeByLocalVariances.cpp:16-19, :151-154).
CV_EXPORTS void binarizeByLocalVariances(cv::Mat& inputImage, cv::Mat& outputImage, double varianceThresholdCoeff = 0.125,
                              int minResultVariance = 25, double gamma = 2.0);
CV_EXPORTS void binarizeByLocalVariancesWithoutFilters(cv::Mat& inputImage, cv::Mat& outputImage, double varianceThresholdCoeff = 0.125,
                                            int minResultVariance = 10);

// SURVEY.md §8f rank 3 — src/backgroundNormalization.h:40.  8UC1 -> 8UC1; 8UC3 / 8UC4 -> 8UC3 (the reference's
// Leptonica round trip drops a fourth channel, src/formatConvert.cpp:193-206).  std::invalid_argument for an empty image
// (src/backgroundNormalization.cpp:40-43).
CV_EXPORTS void backgroundNormalization(const cv::Mat& inputImage, cv::Mat& outputImage);

// SURVEY.md §8f rank 4a — src/deskew/deskew.h:42, src/rotate.h:39.  deskew: gray -> Otsu -> HoughLinesP angle vote ->
// rotate; the result is max(cols, rows) square when an angle was found (src/rotate.cpp:64-68), a clone otherwise.
// The orientation step (src/deskew/deskew.cpp:238) is a no-op for the page the reference hands it.
CV_EXPORTS bool deskew(const cv::Mat& inputImage, cv::Mat& outputImage);
// src/deskew/deskew.h:62 - the angle prl::deskew rotates by: HoughLinesP on the complement of a 1-channel (thresholded)
// page, first-fit vote over the segments' atan2, in degrees; 0.0 when no segment is found (deskew.cpp:139-205).
CV_EXPORTS double findAngle(const cv::Mat& inputImage);
// src/deskew/deskew.h:52 - 0.0 (see prl_host.cpp: a no-op for the page prl::deskew hands it, deskew.cpp:70-84, :238).
CV_EXPORTS double findOrientation(const cv::Mat& inputImage);
CV_EXPORTS void rotate(const cv::Mat& inputImage, cv::Mat& outputImage, double angle);

// BASELINE config 1 (plumbing, host only): global Otsu, the one global threshold the reference uses
// (cv::threshold(..., THRESH_BINARY | THRESH_OTSU), src/deskew/deskew.cpp:224).  Not a GPU path.
CV_EXPORTS void binarize(cv::Mat& inputImage, cv::Mat& outputImage);

}  // namespace prl
