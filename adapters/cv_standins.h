/*
 * adapters/cv_standins.h -- the few cv:: types the adapters touch, for builds WITHOUT OpenCV (this repo's own test
 * programs).  With OpenCV present the real headers are used and this file defines nothing.
 * cv::Matx33f / Matx31f arithmetic follows OpenCV's generic Matx code (float accumulation in index order, closed-form
 * 3x3 determinant / inverse), so that the geometry the matcher adapter forms (F12 = K1^-T [t12]x R12 K2^-1,
 * src/CameraModels/Pinhole.cpp:161-164) is evaluated with the same operations.
 */
#ifndef ORBFE_ADAPTER_CV_STANDINS_H
#define ORBFE_ADAPTER_CV_STANDINS_H

#include <cstddef>
#include <cstdint>
#include <vector>

#if !defined(ORBFE_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define ORBFE_HAVE_OPENCV 1
#endif
#endif

#ifndef ORBFE_HAVE_OPENCV
namespace cv {
struct Point2f {
    float x, y;
};
struct KeyPoint { // same layout as cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
#ifndef CV_8U
#define CV_8U 0
#define CV_32F 5
#endif
class Mat { // rows x cols of CV_8U (images, descriptors) or CV_32F (poses) with a row step; just enough for the adapters
public:
    int rows = 0, cols = 0;
    size_t step = 0; // bytes
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type = CV_8U) { create(r, c, type); }
    Mat(int r, int c, uint8_t* ext, size_t s) : rows(r), cols(c), step(s), data(ext) {}
    Mat(const Mat& o) { *this = o; }
    Mat& operator=(const Mat& o)
    {
        rows = o.rows; cols = o.cols; step = o.step; type_ = o.type_;
        store = o.store;
        data = o.store.empty() ? o.data : store.data(); // owning matrices copy deep, views stay views
        return *this;
    }
    void create(int r, int c, int type = CV_8U)
    {
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * (type == CV_32F ? 4 : 1);
        store.assign((size_t)r * step, 0);
        data = store.data();
    }
    void release()
    {
        rows = cols = 0;
        step = 0;
        store.clear();
        data = nullptr;
    }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    bool isContinuous() const { return step == (size_t)cols * (type_ == CV_32F ? 4 : 1); }
    // views, as cv::Mat::rowRange / colRange give them (what Frame::ComputeStereoMatches takes of a pyramid level)
    Mat rowRange(int r0, int r1) const { return Mat(r1 - r0, cols, data + (size_t)r0 * step, step); }
    Mat colRange(int c0, int c1) const { return Mat(rows, c1 - c0, data + (size_t)c0 * (type_ == CV_32F ? 4 : 1), step); }
    int type() const { return type_; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
    template <class T> T& at(int r, int c = 0) { return reinterpret_cast<T*>(data + (size_t)r * step)[c]; }
    template <class T> const T& at(int r, int c = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step)[c]; }

private:
    int type_ = CV_8U;
    std::vector<uint8_t> store;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
struct Point3f {
    float x, y, z;
    Point3f() : x(0), y(0), z(0) {}
    Point3f(float a, float b, float c) : x(a), y(b), z(c) {}
};
template <int M, int N>
struct MatxF {
    float val[M * N];
    MatxF()
    {
        for (int i = 0; i < M * N; i++) val[i] = 0;
    }
    float& operator()(int i, int j) { return val[i * N + j]; }
    const float& operator()(int i, int j) const { return val[i * N + j]; }
    float& operator()(int i) { return val[i]; }
    const float& operator()(int i) const { return val[i]; }
    MatxF<N, M> t() const
    {
        MatxF<N, M> r;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) r(j, i) = (*this)(i, j);
        return r;
    }
    MatxF<M, N> inv() const; // 3x3 only
};
template <int M, int K, int N>
inline MatxF<M, N> operator*(const MatxF<M, K>& a, const MatxF<K, N>& b)
{
    MatxF<M, N> c;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float s = 0;
            for (int k = 0; k < K; k++) s += a(i, k) * b(k, j);
            c(i, j) = s;
        }
    return c;
}
template <int M, int N>
inline MatxF<M, N> operator+(const MatxF<M, N>& a, const MatxF<M, N>& b)
{
    MatxF<M, N> c;
    for (int i = 0; i < M * N; i++) c.val[i] = a.val[i] + b.val[i];
    return c;
}
template <int M, int N>
inline MatxF<M, N> operator-(const MatxF<M, N>& a)
{
    MatxF<M, N> c;
    for (int i = 0; i < M * N; i++) c.val[i] = -a.val[i];
    return c;
}
template <>
inline MatxF<3, 3> MatxF<3, 3>::inv() const
{
    const MatxF<3, 3>& a = *this;
    MatxF<3, 3> b;
    float d = a(0, 0) * (a(1, 1) * a(2, 2) - a(2, 1) * a(1, 2)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(2, 0) * a(1, 2)) +
              a(0, 2) * (a(1, 0) * a(2, 1) - a(2, 0) * a(1, 1));
    if (d == 0) return b;
    d = 1 / d;
    b(0, 0) = (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) * d;
    b(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * d;
    b(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * d;
    b(1, 0) = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * d;
    b(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * d;
    b(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * d;
    b(2, 0) = (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)) * d;
    b(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * d;
    b(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * d;
    return b;
}
typedef MatxF<3, 3> Matx33f;
typedef MatxF<3, 1> Matx31f;
} // namespace cv
#endif

#endif
