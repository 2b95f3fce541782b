// DECLARATION-ONLY mock of the handful of OpenCV types the adapters' ORBFE_HAVE_OPENCV branches and adapters/diff_opencv.cpp
// touch.  Test infrastructure: lets `g++ -fsyntax-only -I tests/opencv_mock` parse those branches in a container that has
// no OpenCV (tests/test_adapter_syntax.py).  Nothing here is ever linked or shipped; signatures follow OpenCV 4's
// opencv2/core.hpp closely enough for overload resolution, nothing more.
#ifndef ORBFE_OPENCV_MOCK_CORE_HPP
#define ORBFE_OPENCV_MOCK_CORE_HPP
#include <cstddef>
#include <cstdint>
#include <vector>

#define ORBFE_OPENCV_IS_A_MOCK 1
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) ((void)(expr))
typedef unsigned char uchar;

namespace cv {
template <class T>
struct Point_ {
    T x, y;
    Point_();
    Point_(T x_, T y_);
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;
template <class T>
struct Point3_ {
    T x, y, z;
    Point3_();
    Point3_(T x_, T y_, T z_);
};
typedef Point3_<float> Point3f;
struct Rect {
    int x, y, width, height;
    Rect();
    Rect(int x_, int y_, int w, int h);
};
struct Size {
    int width, height;
    Size();
    Size(int w, int h);
};
class KeyPoint {
public:
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint();
};
struct MatStep {
    size_t v;
    operator size_t() const;
};
class Mat {
public:
    int flags, dims, rows, cols;
    uchar* data;
    MatStep step;
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data, size_t step = 0);
    Mat(const Mat&);
    Mat& operator=(const Mat&);
    ~Mat();
    bool empty() const;
    int type() const;
    void create(int rows, int cols, int type);
    void release();
    uchar* ptr(int row = 0);
    const uchar* ptr(int row = 0) const;
    template <class T>
    T& at(int r, int c = 0);
    template <class T>
    const T& at(int r, int c = 0) const;
    Mat operator()(const Rect& roi) const;
    Mat clone() const;
    Mat row(int r) const;
    Mat t() const;
    bool isContinuous() const;
};
class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat&);
    template <class T>
    _InputArray(const std::vector<T>&);
    Mat getMat(int idx = -1) const;
    bool empty() const;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat&);
    template <class T>
    _OutputArray(std::vector<T>&);
    void create(int rows, int cols, int type) const;
    void release() const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
const _InputArray& noArray();

template <class T, int m, int n>
class Matx {
public:
    T val[m * n];
    Matx();
    T& operator()(int i, int j);
    const T& operator()(int i, int j) const;
    T& operator()(int i);
    const T& operator()(int i) const;
    Matx<T, n, m> t() const;
    Matx<T, m, n> inv() const;
};
typedef Matx<float, 3, 3> Matx33f;
typedef Matx<float, 3, 1> Matx31f;
typedef Matx<float, 4, 4> Matx44f;
template <class T, int m, int k, int n>
Matx<T, m, n> operator*(const Matx<T, m, k>&, const Matx<T, k, n>&);
template <class T, int m, int n>
Matx<T, m, n> operator+(const Matx<T, m, n>&, const Matx<T, m, n>&);
template <class T, int m, int n>
Matx<T, m, n> operator-(const Matx<T, m, n>&, const Matx<T, m, n>&);

enum BorderTypes { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4,
                   BORDER_ISOLATED = 16 };
enum NormTypes { NORM_L2 = 4, NORM_HAMMING = 6 };
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
float fastAtan2(float y, float x);
int cvRound(double v);
class SVD {
public:
    enum Flags { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
    static void compute(InputArray src, OutputArray w, OutputArray u, OutputArray vt, int flags = 0);
};
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
};
template <class T>
class Ptr {
public:
    Ptr();
    T* operator->() const;
};
}  // namespace cv
#endif
