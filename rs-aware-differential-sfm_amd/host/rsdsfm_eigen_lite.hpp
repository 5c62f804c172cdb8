// rsdsfm_eigen_lite.hpp -- the handful of Eigen types the reference's solver boundary uses.
// With Eigen on the include path the real types are used; otherwise minimal column-major stand-ins with the same
// storage layout (so .data() can be handed to the C ABI) and the accessors the call sites need.
#pragma once

#if __has_include(<Eigen/Dense>) && !defined(RSDSFM_FORCE_LITE)
#include <Eigen/Dense>
namespace rsdsfm {
namespace lite = Eigen;
}
#else
#include <cstddef>
#include <ostream>
#include <vector>

namespace rsdsfm {
namespace lite {

// column-major Rows x N (Rows fixed or dynamic), contiguous storage
template <int Rows>
class ArrayRX {
public:
    ArrayRX() : cols_(0) {}
    ArrayRX(long rows, long cols) : cols_(cols), d_((size_t)(Rows * cols), 0.0) { (void)rows; }
    static ArrayRX Zero(long rows, long cols) { return ArrayRX(rows, cols); }
    static ArrayRX Ones(long rows, long cols) {
        ArrayRX a(rows, cols);
        for (double& x : a.d_) x = 1.0;
        return a;
    }
    long rows() const { return Rows; }
    long cols() const { return cols_; }
    double& operator()(long r, long c) { return d_[(size_t)(c * Rows + r)]; }
    double operator()(long r, long c) const { return d_[(size_t)(c * Rows + r)]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    void conservativeResize(long rows, long cols) {
        (void)rows;
        d_.resize((size_t)(Rows * cols));
        cols_ = cols;
    }

private:
    long cols_;
    std::vector<double> d_;
};
using Array2Xd = ArrayRX<2>;
using Array3Xd = ArrayRX<3>;
using Matrix2Xd = ArrayRX<2>;

class ArrayXd {
public:
    ArrayXd() {}
    explicit ArrayXd(long n) : d_((size_t)n, 0.0) {}
    static ArrayXd Zero(long n) { return ArrayXd(n); }
    static ArrayXd Ones(long n) {
        ArrayXd a(n);
        for (double& x : a.d_) x = 1.0;
        return a;
    }
    // coefficient-wise scalar updates (the reference's global-shutter switch: `alpha *= 0; alpha += 1;`, main.cc:441-444)
    ArrayXd& operator*=(double s) {
        for (double& x : d_) x *= s;
        return *this;
    }
    ArrayXd& operator+=(double s) {
        for (double& x : d_) x += s;
        return *this;
    }
    long size() const { return (long)d_.size(); }
    long rows() const { return (long)d_.size(); }
    double& operator()(long i) { return d_[(size_t)i]; }
    double operator()(long i) const { return d_[(size_t)i]; }
    double& operator[](long i) { return d_[(size_t)i]; }
    double operator[](long i) const { return d_[(size_t)i]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    void conservativeResize(long n) { d_.resize((size_t)n); }

private:
    std::vector<double> d_;
};
using VectorXd = ArrayXd;

class Vector3d {
public:
    Vector3d() : d_{0, 0, 0} {}
    Vector3d(double x, double y, double z) : d_{x, y, z} {}
    static Vector3d Zero() { return Vector3d(); }
    double& operator()(int i) { return d_[i]; }
    double operator()(int i) const { return d_[i]; }
    double& operator[](int i) { return d_[i]; }
    double operator[](int i) const { return d_[i]; }
    double x() const { return d_[0]; }
    double y() const { return d_[1]; }
    double z() const { return d_[2]; }
    double* data() { return d_; }
    const double* data() const { return d_; }
    Vector3d& operator*=(double s) {
        d_[0] *= s, d_[1] *= s, d_[2] *= s;
        return *this;
    }
    // what `std::cout << w.transpose()` needs (main.cc:449-451): a row view that streams as "x y z"
    struct RowView {
        double x, y, z;
    };
    RowView transpose() const { return RowView{d_[0], d_[1], d_[2]}; }

private:
    double d_[3];
};

inline std::ostream& operator<<(std::ostream& os, const Vector3d::RowView& r) { return os << r.x << " " << r.y << " " << r.z; }

class Vector2d {
public:
    Vector2d() : d_{0, 0} {}
    Vector2d(double x, double y) : d_{x, y} {}
    static Vector2d Zero() { return Vector2d(); }
    double& operator()(int i) { return d_[i]; }
    double operator()(int i) const { return d_[i]; }
    double& operator[](int i) { return d_[i]; }
    double operator[](int i) const { return d_[i]; }
    double x() const { return d_[0]; }
    double y() const { return d_[1]; }
    double* data() { return d_; }
    const double* data() const { return d_; }

private:
    double d_[2];
};

class Vector2i {
public:
    Vector2i() : d_{0, 0} {}
    Vector2i(int x, int y) : d_{x, y} {}
    int& operator()(int i) { return d_[i]; }
    int operator()(int i) const { return d_[i]; }
    int x() const { return d_[0]; }
    int y() const { return d_[1]; }

private:
    int d_[2];
};

class Matrix3d {  // column-major like Eigen
public:
    Matrix3d() : d_{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
    static Matrix3d Zero() { return Matrix3d(); }
    static Matrix3d Identity() {
        Matrix3d m;
        m(0, 0) = m(1, 1) = m(2, 2) = 1.0;
        return m;
    }
    double& operator()(int r, int c) { return d_[c * 3 + r]; }
    double operator()(int r, int c) const { return d_[c * 3 + r]; }
    double* data() { return d_; }
    const double* data() const { return d_; }

private:
    double d_[9];
};

class MatrixXd {  // column-major
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(long r, long c) : r_(r), c_(c), d_((size_t)(r * c), 0.0) {}
    static MatrixXd Zero(long r, long c) { return MatrixXd(r, c); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    double& operator()(long r, long c) { return d_[(size_t)(c * r_ + r)]; }
    double operator()(long r, long c) const { return d_[(size_t)(c * r_ + r)]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }

private:
    long r_, c_;
    std::vector<double> d_;
};

}  // namespace lite
}  // namespace rsdsfm
#endif

#include <vector>

namespace rsdsfm {

// stand-in for the reference's cv::Mat of type CV_8UC3: row-major rows x cols x (b, g, r)
class ImageBGR {
public:
    ImageBGR() : r_(0), c_(0) {}
    ImageBGR(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * (size_t)cols * 3, 0) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    unsigned char* data() { return d_.data(); }
    const unsigned char* data() const { return d_.data(); }
    unsigned char& at(int row, int col, int ch) { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 3 + (size_t)ch]; }
    unsigned char at(int row, int col, int ch) const { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 3 + (size_t)ch]; }
    ImageBGR clone() const { return *this; }

private:
    int r_, c_;
    std::vector<unsigned char> d_;
};

// stand-in for cv::Mat_<cv::Vec3f>: row-major rows x cols x 3 floats
class ImageXYZf {
public:
    ImageXYZf() : r_(0), c_(0) {}
    ImageXYZf(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * (size_t)cols * 3, 0.0f) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    float* data() { return d_.data(); }
    const float* data() const { return d_.data(); }
    float at(int row, int col, int ch) const { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 3 + (size_t)ch]; }
    float& at(int row, int col, int ch) { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 3 + (size_t)ch]; }

private:
    int r_, c_;
    std::vector<float> d_;
};

// stand-in for cv::Mat_<cv::Point_<double>>: row-major rows x cols x (dx, dy)
class FlowImage {
public:
    FlowImage() : r_(0), c_(0) {}
    FlowImage(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * (size_t)cols * 2, 0.0) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    double x(int row, int col) const { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 2]; }
    double y(int row, int col) const { return d_[((size_t)row * (size_t)c_ + (size_t)col) * 2 + 1]; }

private:
    int r_, c_;
    std::vector<double> d_;
};

}  // namespace rsdsfm

