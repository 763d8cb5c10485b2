// pnec_types.h -- minimal POD linear-algebra types for the host facade.
//
// The reference's public signatures use Eigen / Sophus / opengv types (Eigen::Vector3d,
// Eigen::Matrix3d, Eigen::Quaterniond, Sophus::SE3d, opengv::bearingVectors_t).  None of those
// libraries exist in this image, so the facade carries layout-compatible stand-ins: a
// std::vector<pnec::Vector3d> is the same 24-byte-stride array as std::vector<Eigen::Vector3d>,
// pnec::Matrix3d is column-major like Eigen::Matrix3d, pnec::Quaterniond stores x,y,z,w like
// Eigen's coeffs().  A build that has Eigen can reinterpret_cast between them.
#pragma once

#include <array>
#include <cmath>
#include <vector>

namespace pnec {

struct Vector3d {
  double v[3] = {0.0, 0.0, 0.0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
  double dot(const Vector3d &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  Vector3d cross(const Vector3d &o) const {
    return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
  }
  Vector3d normalized() const {
    const double n = norm();
    return {v[0] / n, v[1] / n, v[2] / n};
  }
  Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
  Vector3d operator+(const Vector3d &o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
  Vector3d operator-(const Vector3d &o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
  Vector3d operator-() const { return {-v[0], -v[1], -v[2]}; }
  const double *data() const { return v; }
  double *data() { return v; }
};

// column-major 3x3 (element (r,c) at m[3*c + r]), like Eigen::Matrix3d
struct Matrix3d {
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double &operator()(int r, int c) { return m[3 * c + r]; }
  double operator()(int r, int c) const { return m[3 * c + r]; }
  static Matrix3d Identity() {
    Matrix3d I;
    I(0, 0) = I(1, 1) = I(2, 2) = 1.0;
    return I;
  }
  static Matrix3d Zero() { return Matrix3d(); }
  Matrix3d transpose() const {
    Matrix3d t;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) t(r, c) = (*this)(c, r);
    return t;
  }
  Matrix3d operator*(const Matrix3d &o) const {
    Matrix3d p;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += (*this)(r, k) * o(k, c);
        p(r, c) = a;
      }
    return p;
  }
  Vector3d operator*(const Vector3d &x) const {
    Vector3d y;
    for (int r = 0; r < 3; ++r) y[r] = (*this)(r, 0) * x[0] + (*this)(r, 1) * x[1] + (*this)(r, 2) * x[2];
    return y;
  }
  Matrix3d operator*(double s) const {
    Matrix3d p;
    for (int i = 0; i < 9; ++i) p.m[i] = m[i] * s;
    return p;
  }
  const double *data() const { return m; }
  double *data() { return m; }
};

// storage x,y,z,w (Eigen::Quaterniond::coeffs()); constructor order w,x,y,z like Eigen
struct Quaterniond {
  double c[4] = {0.0, 0.0, 0.0, 1.0};
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : c{x, y, z, w} {}
  explicit Quaterniond(const Matrix3d &R);  // Shepperd, as Eigen does
  double x() const { return c[0]; }
  double y() const { return c[1]; }
  double z() const { return c[2]; }
  double w() const { return c[3]; }
  const double *coeffs() const { return c; }
  double *coeffs() { return c; }
  double norm() const { return std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]); }
  Quaterniond normalized() const {
    const double n = norm();
    return Quaterniond(c[3] / n, c[0] / n, c[1] / n, c[2] / n);
  }
  Matrix3d toRotationMatrix() const;  // no normalisation, as Eigen
};

using Matrix4d = std::array<double, 16>;  // row-major 4x4 (pybind boundary only)

// rigid transform: what the path needs of Sophus::SE3d
struct SE3d {
  Matrix3d R = Matrix3d::Identity();
  Vector3d t;
  SE3d() = default;
  SE3d(const Matrix3d &rotation, const Vector3d &translation) : R(rotation), t(translation) {}
  SE3d(const Quaterniond &q, const Vector3d &translation)
      : R(q.normalized().toRotationMatrix()), t(translation) {}
  const Matrix3d &rotationMatrix() const { return R; }
  const Vector3d &translation() const { return t; }
  Vector3d &translation() { return t; }
  Quaterniond unit_quaternion() const { return Quaterniond(R).normalized(); }
  Matrix4d matrix() const {
    Matrix4d M{};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) M[4 * r + c] = R(r, c);
      M[4 * r + 3] = t[r];
    }
    M[15] = 1.0;
    return M;
  }
};

using bearingVectors_t = std::vector<Vector3d>;  // opengv::bearingVectors_t stand-in

inline Quaterniond::Quaterniond(const Matrix3d &R) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0.0) {
    double s = std::sqrt(tr + 1.0);
    c[3] = 0.5 * s;
    s = 0.5 / s;
    c[0] = (R(2, 1) - R(1, 2)) * s;
    c[1] = (R(0, 2) - R(2, 0)) * s;
    c[2] = (R(1, 0) - R(0, 1)) * s;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    c[i] = 0.5 * s;
    s = 0.5 / s;
    c[3] = (R(k, j) - R(j, k)) * s;
    c[j] = (R(j, i) + R(i, j)) * s;
    c[k] = (R(k, i) + R(i, k)) * s;
  }
}

inline Matrix3d Quaterniond::toRotationMatrix() const {
  const double x = c[0], y = c[1], z = c[2], w = c[3];
  Matrix3d R;
  R(0, 0) = 1.0 - 2.0 * (y * y + z * z); R(0, 1) = 2.0 * (x * y - w * z); R(0, 2) = 2.0 * (x * z + w * y);
  R(1, 0) = 2.0 * (x * y + w * z); R(1, 1) = 1.0 - 2.0 * (x * x + z * z); R(1, 2) = 2.0 * (y * z - w * x);
  R(2, 0) = 2.0 * (x * z - w * y); R(2, 1) = 2.0 * (y * z + w * x); R(2, 2) = 1.0 - 2.0 * (x * x + y * y);
  return R;
}

}  // namespace pnec
