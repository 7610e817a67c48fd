// tests/cpp/opencv_api/opencv2/imgproc/imgproc.hpp — DECLARATION-ONLY compile-conformance header (see ../core/core.hpp).
// The one imgproc function the host layer calls when OpenCV is present: cv::cvtColor, with the signature and the
// conversion codes of opencv2/imgproc.hpp (the values of enum ColorConversionCodes have been stable since 2.4).
#ifndef PRL_TEST_OPENCV_API_IMGPROC_HPP
#define PRL_TEST_OPENCV_API_IMGPROC_HPP

#include "../core/core.hpp"

namespace cv {

enum ColorConversionCodes {
    COLOR_BGR2BGRA = 0,
    COLOR_BGRA2BGR = 1,
    COLOR_BGR2RGBA = 2,
    COLOR_RGBA2BGR = 3,
    COLOR_BGR2RGB = 4,
    COLOR_BGRA2RGBA = 5,
    COLOR_BGR2GRAY = 6,
    COLOR_RGB2GRAY = 7,
    COLOR_GRAY2BGR = 8,
    COLOR_GRAY2BGRA = 9,
    COLOR_BGRA2GRAY = 10,
    COLOR_RGBA2GRAY = 11,
};

CV_EXPORTS void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);

}  // namespace cv

#endif  // PRL_TEST_OPENCV_API_IMGPROC_HPP
