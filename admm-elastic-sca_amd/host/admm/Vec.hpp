// Vec.hpp -- the vector types of the host-side mirror.
//
// The reference's public members and virtual signatures use Eigen::VectorXd /
// Eigen::Vector3d / Eigen::Triplet<double>
// (deps/admm-elastic-sca/src/system/System.hpp:47-49, Force.hpp:46-52,
// ExplicitForce.hpp:57, AnchorForce.hpp:75).  When Eigen is on the include path
// the mirror uses the real types, so reference-side callers (src/ForceBuilder.*,
// src/SimContext.*, samples/*) and user-written Force subclasses compile
// unchanged; otherwise it supplies the subset of the interface those callers
// use: size(), operator[], operator(), resize, conservativeResize, fill,
// setZero, data, 3-vector arithmetic, Triplet(row, col, value).
#pragma once

#if defined(ADMM_HOST_USE_EIGEN) || (__has_include(<Eigen/Dense>) && !defined(ADMM_HOST_NO_EIGEN))
#include <Eigen/Dense>
#include <Eigen/Sparse>
namespace admm {
typedef Eigen::VectorXd VectorXd;
typedef Eigen::Vector3d Vector3d;
}
#else
#include <cmath>
#include <cstddef>
#include <vector>
namespace Eigen {   // same spelling as the reference's callers use
class VectorXd {
public:
    VectorXd() {}
    explicit VectorXd(std::ptrdiff_t n) : d_(n) {}
    std::ptrdiff_t size() const { return (std::ptrdiff_t)d_.size(); }
    double &operator[](std::ptrdiff_t i) { return d_[i]; }
    const double &operator[](std::ptrdiff_t i) const { return d_[i]; }
    double &operator()(std::ptrdiff_t i) { return d_[i]; }
    const double &operator()(std::ptrdiff_t i) const { return d_[i]; }
    void resize(std::ptrdiff_t n) { d_.assign(n, 0.0); }
    void conservativeResize(std::ptrdiff_t n) { d_.resize(n); }
    void fill(double v) { for (double &x : d_) x = v; }
    void setZero() { fill(0.0); }
    double *data() { return d_.data(); }
    const double *data() const { return d_.data(); }
private:
    std::vector<double> d_;
};
class Vector3d {
public:
    Vector3d() { v_[0] = v_[1] = v_[2] = 0.0; }
    Vector3d(double x, double y, double z) { v_[0] = x; v_[1] = y; v_[2] = z; }
    double &operator[](int i) { return v_[i]; }
    const double &operator[](int i) const { return v_[i]; }
    double &operator()(int i) { return v_[i]; }
    const double &operator()(int i) const { return v_[i]; }
    void setZero() { v_[0] = v_[1] = v_[2] = 0.0; }
    Vector3d operator-(const Vector3d &o) const { return Vector3d(v_[0] - o.v_[0], v_[1] - o.v_[1], v_[2] - o.v_[2]); }
    Vector3d operator+(const Vector3d &o) const { return Vector3d(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2]); }
    Vector3d operator*(double s) const { return Vector3d(v_[0] * s, v_[1] * s, v_[2] * s); }
    Vector3d operator/(double s) const { return Vector3d(v_[0] / s, v_[1] / s, v_[2] / s); }
    Vector3d &operator+=(const Vector3d &o) { v_[0] += o.v_[0]; v_[1] += o.v_[1]; v_[2] += o.v_[2]; return *this; }
    Vector3d &operator-=(const Vector3d &o) { v_[0] -= o.v_[0]; v_[1] -= o.v_[1]; v_[2] -= o.v_[2]; return *this; }
    friend Vector3d operator*(double s, const Vector3d &a) { return a * s; }
    double dot(const Vector3d &o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
    double norm() const { return std::sqrt(dot(*this)); }
private:
    double v_[3];
};
template <class T> class Triplet {
public:
    Triplet() : r_(0), c_(0), v_(0) {}
    Triplet(int r, int c, const T &v = T(0)) : r_(r), c_(c), v_(v) {}
    int row() const { return r_; }
    int col() const { return c_; }
    const T &value() const { return v_; }
private:
    int r_, c_;
    T v_;
};
} // namespace Eigen
namespace admm {
typedef Eigen::VectorXd VectorXd;
typedef Eigen::Vector3d Vector3d;
}
#endif
