// cvmat_shim.h — the few pieces of cv::Mat the PRLib hot path touches, for builds without OpenCV.
//
// prl.h includes <opencv2/core/core.hpp> when it exists and this file otherwise (neither the build
// container nor the GPU box has OpenCV).  Only what prl::binarize*/prl::denoise and their callers
// (samples/binarizations/binarizeSauvola_sample.cpp:48-53) use is provided: a ref-counted 8-bit
// matrix header with rows/cols/step/data, create/clone/copyTo, ROI views, and the CV_8UCn type codes.
//
// The shim is a STRICT SUBSET of OpenCV's API: every name it offers exists in <opencv2/core/core.hpp> with the same
// signature (cv::Exception's five-argument constructor, cv::Error codes, a global `uchar`, Mat's constructors and members),
// so code that compiles against it compiles against OpenCV.  tests/test_cpp_host.py holds it to that by compiling the
// host layer and its callers against tests/cpp/opencv_api/ (declaration-only headers with OpenCV's signatures).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <exception>
#include <string>

#define CV_8U 0
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC2 CV_MAKETYPE(CV_8U, 2)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define PRL_CVMAT_SHIM 1

typedef unsigned char uchar;   // global, as in opencv2/core/hal/interface.h

namespace cv {

typedef std::string String;   // opencv2/core/cvstd.hpp (4.x)

namespace Error {
enum Code {   // opencv2/core/base.hpp
    StsOk = 0,
    StsError = -2,
    StsNoMem = -4,
    StsBadArg = -5,
    StsUnsupportedFormat = -210,
    StsOutOfRange = -211,
    StsNotImplemented = -213,
    StsAssert = -215,
    GpuApiCallError = -217,
};
}  // namespace Error

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

// opencv2/core.hpp: class Exception : public std::exception, with exactly this constructor and these members
class Exception : public std::exception {
public:
    Exception() : code(0), line(0) {}
    Exception(int _code, const String& _err, const String& _func, const String& _file, int _line)
        : code(_code), err(_err), func(_func), file(_file), line(_line)
    {
        formatMessage();
    }
    virtual ~Exception() throw() {}
    virtual const char* what() const throw() { return msg.c_str(); }
    void formatMessage()   // the text cv::Exception::formatMessage builds (the version prefix left out)
    {
        msg = file + ":" + std::to_string(line) + ": error: (" + std::to_string(code) + ") " + err;
        if (!func.empty()) msg += " in function '" + func + "'";
        msg += "\n";
    }
    String msg;
    int code;
    String err, func, file;
    int line;
};

class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    // user-allocated data (not owned), as cv::Mat(rows, cols, type, void* data, size_t step)
    Mat(int r, int c, int type, void* d, size_t st = 0)
        : rows(r), cols(c), data(static_cast<uchar*>(d)), type_(type)
    {
        step = st ? st : (size_t)c * channels();
    }

    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && data) return;   // as cv::Mat::create: a matching header (a ROI view too) is kept
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * channels();
        const size_t bytes = step * (size_t)r;
        owner_.reset(bytes ? new uchar[bytes] : nullptr, std::default_delete<uchar[]>());
        data = owner_.get();
    }

    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return cn(type_); }
    size_t elemSize() const { return (size_t)channels(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * channels(); }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }

    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T& at(int y, int x) { return reinterpret_cast<T*>(data + (size_t)y * step)[x]; }
    template <typename T> const T& at(int y, int x) const { return reinterpret_cast<const T*>(data + (size_t)y * step)[x]; }

    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat& m) const
    {
        m.create(rows, cols, type_);
        for (int y = 0; y < rows; ++y) std::memcpy(m.ptr(y), ptr(y), (size_t)cols * channels());
    }
    Mat operator()(const Rect& r) const
    {
        if (r.x < 0 || r.y < 0 || r.width < 0 || r.height < 0 || r.x + r.width > cols || r.y + r.height > rows)
            throw Exception(Error::StsAssert, "0 <= roi.x && 0 <= roi.width && roi.x + roi.width <= m.cols && 0 <= roi.y && 0 <= roi.height && roi.y + roi.height <= m.rows",
                            "Mat", __FILE__, __LINE__);
        Mat m(*this);
        m.data = data + (size_t)r.y * step + (size_t)r.x * channels();
        m.rows = r.height;
        m.cols = r.width;
        return m;
    }
    void release()
    {
        owner_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }

private:
    static int cn(int type) { return (type >> CV_CN_SHIFT) + 1; }
    int type_ = CV_8UC1;
    std::shared_ptr<uchar> owner_;
};

}  // namespace cv
