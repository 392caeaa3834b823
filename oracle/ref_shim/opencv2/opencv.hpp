// opencv2/opencv.hpp -- STAND-IN, not OpenCV.  TEST INFRASTRUCTURE (oracle/_ref build only).
//
// Just enough of the `cv` API for the reference's OWN sources (code/esac/esac.cpp, esac_types.h, esac_util.h,
// esac_loss.h, esac_derivative.h) to compile unmodified from /root/reference, so that the reference's control
// flow (sampling loops, float/double mixes, traversal orders, refinement stopping rule, gradient assembly,
// clamps, the loss and its derivative) runs as written.  The numerical routines OpenCV would provide
// (solvePnP P3P / ITERATIVE, projectPoints, Rodrigues, matrix inverses) are the oracle's restatements
// (oracle/esac_oracle.c) -- they stay "from memory"; what this build pins is everything AROUND them.
// Matrix arithmetic is plain double loops in index order (cv::gemm's exact summation order is not modelled).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <vector>

#include "../../esac_oracle.h"

typedef unsigned char uchar;  // OpenCV puts uchar in the global namespace (cvdef.h)

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6

namespace cv {

using ::uchar;
enum { SOLVEPNP_ITERATIVE = 0, SOLVEPNP_EPNP = 1, SOLVEPNP_P3P = 2 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
};
template <typename T> Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point3_<float> Point3f;
// cv::norm(Point_<T>) accumulates in double: sqrt((double)x*x + (double)y*y)
template <typename T> double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double v[4];
    double operator[](int i) const { return v[i]; }
};

template <typename T> struct DataType { enum { type = 64 + sizeof(T) }; };  // structs (Point2i ...)
template <> struct DataType<uchar> { enum { type = CV_8U }; };
template <> struct DataType<int> { enum { type = CV_32S }; };
template <> struct DataType<float> { enum { type = CV_32F }; };
template <> struct DataType<double> { enum { type = CV_64F }; };

template <typename T> class Mat_;

// Untyped matrix header over shared storage (views share the buffer, like cv::Mat).
class Mat {
public:
    int rows, cols;
    Mat() : rows(0), cols(0), type_(0), esz_(0), step_(0), off_(0) {}
    Mat(int r, int c, int type, size_t esz) : rows(r), cols(c), type_(type), esz_(esz), step_((size_t)c * esz), off_(0) {
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * esz, 0);
    }
    explicit Mat(const Point3f& p) : Mat(3, 1, CV_32F, sizeof(float)) {  // cv::Mat(obj): 3x1 float
        at<float>(0, 0) = p.x; at<float>(1, 0) = p.y; at<float>(2, 0) = p.z;
    }
    int type() const { return empty() ? 0 : type_; }
    bool empty() const { return rows == 0 || cols == 0 || !buf_; }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }
    unsigned char* ptr(int y, int x) const { return buf_->data() + off_ + (size_t)y * step_ + (size_t)x * esz_; }
    template <typename T> T& at(int y, int x) { assert((int)DataType<T>::type == type_); return *reinterpret_cast<T*>(ptr(y, x)); }
    template <typename T> const T& at(int y, int x) const { assert((int)DataType<T>::type == type_); return *reinterpret_cast<const T*>(ptr(y, x)); }
    Mat clone() const {
        Mat m(rows, cols, type_, esz_);
        for (int y = 0; y < rows; y++)
            if (cols) std::memcpy(m.ptr(y, 0), ptr(y, 0), (size_t)cols * esz_);
        return m;
    }
    Mat rowRange(int a, int b) const { Mat m = *this; m.rows = b - a; m.off_ = off_ + (size_t)a * step_; return m; }
    Mat colRange(int a, int b) const { Mat m = *this; m.cols = b - a; m.off_ = off_ + (size_t)a * esz_; return m; }
    Mat row(int y) const { return rowRange(y, y + 1); }
    Mat col(int x) const { return colRange(x, x + 1); }
    void copyTo(Mat dst) const {  // dst is a header sharing its owner's buffer: writes land in the owner
        if (dst.empty() || dst.rows != rows || dst.cols != cols) { assert(false && "copyTo: shim needs a pre-sized destination"); return; }
        assert(dst.type_ == type_);
        for (int y = 0; y < rows; y++) std::memcpy(dst.ptr(y, 0), ptr(y, 0), (size_t)cols * esz_);
    }
    double getd(int y, int x) const {  // value as double whatever the element type
        if (type_ == CV_64F) return *reinterpret_cast<const double*>(ptr(y, x));
        if (type_ == CV_32F) return (double)*reinterpret_cast<const float*>(ptr(y, x));
        if (type_ == CV_32S) return (double)*reinterpret_cast<const int*>(ptr(y, x));
        if (type_ == CV_8U) return (double)*ptr(y, x);
        assert(false);
        return 0;
    }
    void convertTo(Mat& dst, int rtype) const;
    Mat t() const;
    Mat inv(int method = DECOMP_LU) const;
protected:
    int type_;
    size_t esz_, step_, off_;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

template <typename T> class Mat_ : public Mat {
public:
    Mat_() : Mat() { type_ = DataType<T>::type; esz_ = sizeof(T); }
    Mat_(int r, int c) : Mat(r, c, DataType<T>::type, sizeof(T)) {}
    Mat_(const Mat& m) : Mat(m) {
        if (m.empty()) { type_ = DataType<T>::type; esz_ = sizeof(T); return; }
        if (m.type() != (int)DataType<T>::type) {  // converting construction (cv::Mat_<double> x = floatMat)
            Mat_ c(m.rows, m.cols);
            for (int y = 0; y < m.rows; y++)
                for (int x = 0; x < m.cols; x++) c(y, x) = (T)m.getd(y, x);
            *this = c;
        }
    }
    Mat_& operator=(const Mat_& o) = default;
    Mat_(const Mat_& o) = default;
    Mat_& operator=(const T& s) {  // OpenCV: sets every element
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) (*this)(y, x) = s;
        return *this;
    }
    T& operator()(int y, int x) { return *reinterpret_cast<T*>(ptr(y, x)); }
    const T& operator()(int y, int x) const { return *reinterpret_cast<const T*>(ptr(y, x)); }
    T& operator()(int i) { return rows == 1 ? (*this)(0, i) : (*this)(i, 0); }  // single index on a vector
    const T& operator()(int i) const { return rows == 1 ? (*this)(0, i) : (*this)(i, 0); }
    static Mat_ zeros(int r, int c) { return Mat_(r, c); }
    static Mat_ zeros(Size s) { return Mat_(s.height, s.width); }
    static Mat_ eye(int r, int c) { Mat_ m(r, c); for (int i = 0; i < r && i < c; i++) m(i, i) = (T)1; return m; }
    Mat_ clone() const { return Mat_(Mat::clone()); }
    Mat_ rowRange(int a, int b) const { return Mat_(Mat::rowRange(a, b)); }
    Mat_ colRange(int a, int b) const { return Mat_(Mat::colRange(a, b)); }
    Mat_ row(int y) const { return Mat_(Mat::row(y)); }
    Mat_ col(int x) const { return Mat_(Mat::col(x)); }
    Mat_ t() const { return Mat_(Mat::t()); }
    Mat_ inv(int method = DECOMP_LU) const { return Mat_(Mat::inv(method)); }
    // in-place ops work on views too (the view shares its owner's storage)
    Mat_ operator*=(double s) { for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) (*this)(y, x) = (T)((*this)(y, x) * s); return *this; }
    Mat_ operator+=(const Mat& o) { assert(o.rows == rows && o.cols == cols); for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) (*this)(y, x) = (T)((*this)(y, x) + o.getd(y, x)); return *this; }
};

inline void Mat::convertTo(Mat& dst, int rtype) const {
    assert(rtype == CV_64F);
    Mat_<double> d(rows, cols);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) d(y, x) = getd(y, x);
    dst = d;
}
inline Mat Mat::t() const {
    Mat_<double> r(cols, rows);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) r(x, y) = getd(y, x);
    return r;
}
inline Mat Mat::inv(int method) const {
    assert(rows == cols);
    if (rows == 4) {  // pose2trans / trans2pose: generic LU (cv::Mat::inv default)
        assert(method == DECOMP_LU);
        double A[16], Ai[16];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) A[4 * i + j] = getd(i, j);
        esac_oracle_inv4(A, Ai);
        Mat_<double> r(4, 4);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r(i, j) = Ai[4 * i + j];
        return r;
    }
    assert(rows == 6 && method == DECOMP_SVD);  // (J^T J).inv(cv::DECOMP_SVD), esac.cpp:434
    double A[36], Ai[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) A[6 * i + j] = getd(i, j);
    esac_oracle_pinv_sym6(A, Ai);
    Mat_<double> r(6, 6);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) r(i, j) = Ai[6 * i + j];
    return r;
}

// ---- arithmetic on Mat (always evaluated in double, results CV_64F) ---------------------------------
inline Mat_<double> operator*(const Mat& a, const Mat& b) {
    assert(a.cols == b.rows);
    Mat_<double> c(a.rows, b.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += a.getd(i, k) * b.getd(k, j);
            c(i, j) = s;
        }
    return c;
}
inline Mat_<double> operator*(const Mat& a, double s) {
    Mat_<double> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = a.getd(i, j) * s;
    return c;
}
inline Mat_<double> operator*(double s, const Mat& a) { return a * s; }
inline Mat_<double> operator/(const Mat& a, double s) {
    Mat_<double> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = a.getd(i, j) / s;
    return c;
}
inline Mat_<double> operator+(const Mat& a, const Mat& b) {
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat_<double> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = a.getd(i, j) + b.getd(i, j);
    return c;
}
inline Mat_<double> operator-(const Mat& a, const Mat& b) {
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat_<double> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = a.getd(i, j) - b.getd(i, j);
    return c;
}
inline Mat_<double> operator-(const Mat& a) {
    Mat_<double> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = -a.getd(i, j);
    return c;
}
inline Mat_<uchar> operator!=(const Mat& a, const Mat& b) {
    Mat_<uchar> c(a.rows, a.cols);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c(i, j) = (a.getd(i, j) != b.getd(i, j)) ? 255 : 0;
    return c;
}
inline Scalar sum(const Mat& a) {
    Scalar s = {{0, 0, 0, 0}};
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) s.v[0] += a.getd(i, j);
    return s;
}
inline Scalar trace(const Mat& a) {
    Scalar s = {{0, 0, 0, 0}};
    for (int i = 0; i < a.rows && i < a.cols; i++) s.v[0] += a.getd(i, i);
    return s;
}
inline double norm(const Mat& a) {
    double s = 0;
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) s += a.getd(i, j) * a.getd(i, j);
    return std::sqrt(s);
}

// ---- calib3d stand-ins: thin adapters onto the oracle's restatements ------------------------------------
inline void read_camera(const Mat& K, double& fx, double& fy, double& cx, double& cy) {
    fx = K.getd(0, 0); fy = K.getd(1, 1); cx = K.getd(0, 2); cy = K.getd(1, 2);  // float camMat widened (esac.cpp:93-97)
}
inline void read_vec3(const Mat& v, double out[3]) {
    assert(v.total() == 3);
    for (int i = 0; i < 3; i++) out[i] = v.rows == 3 ? v.getd(i, 0) : v.getd(0, i);
}

inline void Rodrigues(const Mat& src, Mat& dst, Mat& jacobian) {
    assert(src.total() == 3);
    double r[3], R[9], J[27];
    read_vec3(src, r);
    esac_oracle_rodrigues_vec2mat(r, R, J);
    Mat_<double> m(3, 3), j(3, 9);
    for (int i = 0; i < 9; i++) m(i / 3, i % 3) = R[i];
    for (int i = 0; i < 27; i++) j(i / 9, i % 9) = J[i];
    dst = m;
    jacobian = j;
}
inline void Rodrigues(const Mat& src, Mat& dst) {
    if (src.total() == 3) {
        double r[3], R[9];
        read_vec3(src, r);
        esac_oracle_rodrigues_vec2mat(r, R, nullptr);
        Mat_<double> m(3, 3);
        for (int i = 0; i < 9; i++) m(i / 3, i % 3) = R[i];
        dst = m;
    } else {
        assert(src.rows == 3 && src.cols == 3);
        double R[9], r[3];
        for (int i = 0; i < 9; i++) R[i] = src.getd(i / 3, i % 3);
        esac_oracle_rodrigues_mat2vec_svd(R, r);  // OpenCV re-orthonormalises a matrix input (U*Vt)
        Mat_<double> m(3, 1);
        for (int i = 0; i < 3; i++) m(i, 0) = r[i];
        dst = m;
    }
}

inline void projectPoints(const std::vector<Point3f>& pts, const Mat& rvec, const Mat& tvec, const Mat& K, const Mat&,
                          std::vector<Point2f>& out) {
    double r[3], t[3], fx, fy, cx, cy;
    read_vec3(rvec, r); read_vec3(tvec, t); read_camera(K, fx, fy, cx, cy);
    out.resize(pts.size());
    if (pts.empty()) return;
    static_assert(sizeof(Point3f) == 12 && sizeof(Point2f) == 8, "packed points expected");
    esac_oracle_project(r, t, fx, fy, cx, cy, reinterpret_cast<const float*>(pts.data()), (int)pts.size(),
                        reinterpret_cast<float*>(out.data()));
}
// with the Jacobian: 2n x 15, columns 0..2 d/drvec, 3..5 d/dtvec (the remaining OpenCV columns -- focal length,
// principal point, distortion -- are never read by the reference, which keeps colRange(0, 6))
inline void projectPoints(const std::vector<Point3f>& pts, const Mat& rvec, const Mat& tvec, const Mat& K, const Mat& d,
                          std::vector<Point2f>& out, Mat_<double>& jac) {
    projectPoints(pts, rvec, tvec, K, d, out);
    double r[3], t[3], fx, fy, cx, cy;
    read_vec3(rvec, r); read_vec3(tvec, t); read_camera(K, fx, fy, cx, cy);
    jac = Mat_<double>((int)pts.size() * 2, 15);
    std::vector<double> J((size_t)pts.size() * 12);
    if (!pts.empty())
        esac_oracle_project_jac(r, t, fx, fy, cx, cy, reinterpret_cast<const float*>(pts.data()), (int)pts.size(), J.data());
    for (size_t i = 0; i < pts.size(); i++)
        for (int k = 0; k < 6; k++) {
            jac((int)(2 * i), k) = J[i * 12 + k];
            jac((int)(2 * i + 1), k) = J[i * 12 + 6 + k];
        }
}

inline bool solvePnP(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, const Mat& K, const Mat&,
                     Mat& rvec, Mat& tvec, bool useExtrinsicGuess, int flags) {
    double fx, fy, cx, cy;
    read_camera(K, fx, fy, cx, cy);
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    bool ok;
    if (flags == SOLVEPNP_P3P) {
        assert(obj.size() == 4 && img.size() == 4);  // CV_Assert(npoints == 4)
        double o[12], im[8];
        for (int j = 0; j < 4; j++) {
            o[3 * j] = obj[j].x; o[3 * j + 1] = obj[j].y; o[3 * j + 2] = obj[j].z;
            im[2 * j] = img[j].x; im[2 * j + 1] = img[j].y;
        }
        ok = esac_oracle_p3p(o, im, fx, fy, cx, cy, r, t) != 0;
        if (!ok) return false;
    } else {
        assert(flags == SOLVEPNP_ITERATIVE && useExtrinsicGuess);  // the only other use on the path (esac_util.h:426-436)
        double pose[6];
        read_vec3(rvec, pose); read_vec3(tvec, pose + 3);
        esac_oracle_lm_pnp(reinterpret_cast<const float*>(obj.data()), reinterpret_cast<const float*>(img.data()),
                           (int)obj.size(), fx, fy, cx, cy, pose);
        for (int i = 0; i < 3; i++) { r[i] = pose[i]; t[i] = pose[3 + i]; }
        ok = true;  // cvFindExtrinsicCameraParams2 always reports success
    }
    Mat_<double> rv(3, 1), tv(3, 1);
    for (int i = 0; i < 3; i++) { rv(i, 0) = r[i]; tv(i, 0) = t[i]; }
    rvec = rv; tvec = tv;
    return ok;
}

}  // namespace cv
