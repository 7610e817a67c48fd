// tests/cpp/opencv_api/opencv2/core/core.hpp — DECLARATION-ONLY compile-conformance header.  NOT OpenCV, not linkable.
//
// Neither the build container nor the GPU box has OpenCV, so the OpenCV-present branch of the C++ host layer (prl.h's
// PRL_HAVE_OPENCV) could never be compiled; a round-4 review found it did not build against OpenCV's real
// cv::Exception.  This header declares - with the signatures of OpenCV's public API (4.x: opencv2/core/hal/interface.h,
// core/cvdef.h, core/cvstd.hpp, core/base.hpp, core/types.hpp, core/mat.hpp, core.hpp; the 3.x forms of the members used
// here are the same) - exactly the names the host layer and its callers touch, and nothing else.  tests/test_cpp_host.py
// runs `g++ -fsyntax-only -I tests/cpp/opencv_api` over prl_host.cpp, test_prl_host.cpp and test_dropin_sample.cpp: a use
// of anything OpenCV does not have (as the one-string cv::Exception constructor of the old shim) fails that test.
// It has nothing to do with parity: no function here has a body, nothing can run.
//
// Where OpenCV's declaration is a template instance or a proxy class, the real shape is kept (Size_<int>, Rect_<int>,
// MatSize with operator(), MatStep with operator size_t, _InputArray / _OutputArray) so that only code valid against
// those compiles.
#ifndef PRL_TEST_OPENCV_API_CORE_HPP
#define PRL_TEST_OPENCV_API_CORE_HPP

#include <cstddef>
#include <exception>
#include <string>

// ---- opencv2/core/hal/interface.h -------------------------------------------------------------------------------------
typedef unsigned char uchar;
typedef signed char schar;
typedef unsigned short ushort;

#define CV_CN_MAX 512
#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_MAKE_TYPE CV_MAKETYPE
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC2 CV_MAKETYPE(CV_8U, 2)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_8UC(n) CV_MAKETYPE(CV_8U, (n))
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

// ---- opencv2/core/cvdef.h ------------------------------------------------------------------------------------------------
#define CV_EXPORTS __attribute__((visibility("default")))
#define CV_PI 3.1415926535897932384626433832795

namespace cv {

// ---- opencv2/core/cvstd.hpp (4.x) ------------------------------------------------------------------------------------
typedef std::string String;

// ---- opencv2/core/base.hpp -------------------------------------------------------------------------------------------
namespace Error {
enum Code {
    StsOk = 0,
    StsBackTrace = -1,
    StsError = -2,
    StsInternal = -3,
    StsNoMem = -4,
    StsBadArg = -5,
    StsBadFunc = -6,
    StsNoConv = -7,
    StsAutoTrace = -8,
    StsNullPtr = -27,
    StsVecLengthErr = -28,
    StsUnmatchedFormats = -205,
    StsUnmatchedSizes = -209,
    StsUnsupportedFormat = -210,
    StsOutOfRange = -211,
    StsParseError = -212,
    StsNotImplemented = -213,
    StsBadMemBlock = -214,
    StsAssert = -215,
    GpuNotSupported = -216,
    GpuApiCallError = -217,
};
}  // namespace Error

// ---- opencv2/core.hpp ------------------------------------------------------------------------------------------------
class CV_EXPORTS Exception : public std::exception {
public:
    Exception();
    Exception(int _code, const String& _err, const String& _func, const String& _file, int _line);
    virtual ~Exception() throw();
    virtual const char* what() const throw();
    void formatMessage();

    String msg;
    int code;
    String err;
    String func;
    String file;
    int line;
};

// ---- opencv2/core/types.hpp ------------------------------------------------------------------------------------------
template <typename _Tp> class Point_ {
public:
    Point_();
    Point_(_Tp _x, _Tp _y);
    _Tp x, y;
};
typedef Point_<int> Point2i;
typedef Point2i Point;

template <typename _Tp> class Size_ {
public:
    Size_();
    Size_(_Tp _width, _Tp _height);
    _Tp area() const;
    bool empty() const;
    _Tp width, height;
};
typedef Size_<int> Size2i;
typedef Size2i Size;

template <typename _Tp> class Rect_ {
public:
    Rect_();
    Rect_(_Tp _x, _Tp _y, _Tp _width, _Tp _height);
    Rect_(const Point_<_Tp>& org, const Size_<_Tp>& sz);
    Size_<_Tp> size() const;
    _Tp area() const;
    bool empty() const;
    _Tp x, y, width, height;
};
typedef Rect_<int> Rect2i;
typedef Rect2i Rect;

template <typename _Tp> class Scalar_ {
public:
    Scalar_();
    Scalar_(_Tp v0, _Tp v1, _Tp v2 = 0, _Tp v3 = 0);
    Scalar_(_Tp v0);
    _Tp val[4];
};
typedef Scalar_<double> Scalar;

// ---- opencv2/core/mat.hpp --------------------------------------------------------------------------------------------
class Mat;

class CV_EXPORTS _InputArray {
public:
    _InputArray();
    _InputArray(const Mat& m);
    ~_InputArray();
};
class CV_EXPORTS _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat& m);
};
class CV_EXPORTS _InputOutputArray : public _OutputArray {
public:
    _InputOutputArray();
    _InputOutputArray(Mat& m);
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;

struct CV_EXPORTS MatSize {
    explicit MatSize(int* _p);
    int dims() const;
    Size operator()() const;
    const int& operator[](int i) const;
    int& operator[](int i);
    bool operator==(const MatSize& sz) const;
    bool operator!=(const MatSize& sz) const;
    int* p;
};

struct CV_EXPORTS MatStep {
    MatStep();
    explicit MatStep(size_t s);
    const size_t& operator[](int i) const;
    size_t& operator[](int i);
    operator size_t() const;
    MatStep& operator=(size_t s);
    size_t* p;
    size_t buf[2];

protected:
    MatStep& operator=(const MatStep&);
};

class CV_EXPORTS Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(Size size, int type);
    Mat(int rows, int cols, int type, const Scalar& s);
    Mat(Size size, int type, const Scalar& s);
    Mat(const Mat& m);
    Mat(int rows, int cols, int type, void* data, size_t step = AUTO_STEP);
    Mat(Size size, int type, void* data, size_t step = AUTO_STEP);
    Mat(const Mat& m, const Rect& roi);
    ~Mat();
    Mat& operator=(const Mat& m);

    Mat row(int y) const;
    Mat col(int x) const;
    Mat clone() const;
    void copyTo(OutputArray m) const;
    void copyTo(OutputArray m, InputArray mask) const;
    void convertTo(OutputArray m, int rtype, double alpha = 1, double beta = 0) const;
    Mat& operator=(const Scalar& s);
    Mat& setTo(InputArray value, InputArray mask = _InputArray());
    void create(int rows, int cols, int type);
    void create(Size size, int type);
    void release();
    Mat operator()(const Rect& roi) const;

    bool isContinuous() const;
    bool isSubmatrix() const;
    size_t elemSize() const;
    size_t elemSize1() const;
    int type() const;
    int depth() const;
    int channels() const;
    size_t step1(int i = 0) const;
    bool empty() const;
    size_t total() const;

    uchar* ptr(int i0 = 0);
    const uchar* ptr(int i0 = 0) const;
    template <typename _Tp> _Tp* ptr(int i0 = 0);
    template <typename _Tp> const _Tp* ptr(int i0 = 0) const;
    template <typename _Tp> _Tp& at(int row, int col);
    template <typename _Tp> const _Tp& at(int row, int col) const;

    enum { MAGIC_VAL = 0x42FF0000, AUTO_STEP = 0, CONTINUOUS_FLAG = 1 << 14, SUBMATRIX_FLAG = 1 << 15 };

    int flags;
    int dims;
    int rows, cols;
    uchar* data;
    const uchar* datastart;
    const uchar* dataend;
    const uchar* datalimit;
    MatSize size;
    MatStep step;
};

}  // namespace cv

#endif  // PRL_TEST_OPENCV_API_CORE_HPP
