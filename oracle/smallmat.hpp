// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  Parity status: "parity unpinned" against the real
// reference binary (it cannot be built here: no Eigen/PCL/ROS), pinned instead by
// the known-answer tests in tests/ (finite differences, numpy eigh/solve fixtures).
//
// Minimal fixed-size dense double matrices (row-major) standing in for the
// Eigen::Matrix<double,R,C> types the reference uses (tools.hpp:4, voxel_map.hpp:7).
// No expression templates: every operator evaluates eagerly, left to right.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace vso {

template <int R, int C>
struct Mat {
  double a[R * C];
  Mat() { for (int i = 0; i < R * C; i++) a[i] = 0.0; }
  static Mat Zero() { return Mat(); }
  static Mat Identity() {
    Mat m;
    for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = 1.0;
    return m;
  }
  double &operator()(int r, int c) { return a[r * C + c]; }
  const double &operator()(int r, int c) const { return a[r * C + c]; }
  double &operator[](int i) { return a[i]; }
  const double &operator[](int i) const { return a[i]; }
  void setZero() { for (int i = 0; i < R * C; i++) a[i] = 0.0; }
  void setIdentity() { *this = Identity(); }

  Mat operator+(const Mat &o) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] + o.a[i]; return m; }
  Mat operator-(const Mat &o) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] - o.a[i]; return m; }
  Mat operator-() const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = -a[i]; return m; }
  Mat &operator+=(const Mat &o) { for (int i = 0; i < R * C; i++) a[i] += o.a[i]; return *this; }
  Mat &operator-=(const Mat &o) { for (int i = 0; i < R * C; i++) a[i] -= o.a[i]; return *this; }
  Mat operator*(double s) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] * s; return m; }
  Mat operator/(double s) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] / s; return m; }
  Mat &operator*=(double s) { for (int i = 0; i < R * C; i++) a[i] *= s; return *this; }
  Mat &operator/=(double s) { for (int i = 0; i < R * C; i++) a[i] /= s; return *this; }

  Mat<C, R> transpose() const {
    Mat<C, R> t;
    for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) t(c, r) = (*this)(r, c);
    return t;
  }
  template <int BR, int BC>
  Mat<BR, BC> block(int r0, int c0) const {
    Mat<BR, BC> b;
    for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) b(r, c) = (*this)(r0 + r, c0 + c);
    return b;
  }
  template <int BR, int BC>
  void setBlock(int r0, int c0, const Mat<BR, BC> &b) {
    for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) (*this)(r0 + r, c0 + c) = b(r, c);
  }
  template <int BR, int BC>
  void addBlock(int r0, int c0, const Mat<BR, BC> &b) {
    for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) (*this)(r0 + r, c0 + c) += b(r, c);
  }
  Mat<R, 1> col(int c) const { Mat<R, 1> v; for (int r = 0; r < R; r++) v[r] = (*this)(r, c); return v; }
  double squaredNorm() const { double s = 0; for (int i = 0; i < R * C; i++) s += a[i] * a[i]; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  double trace() const { double s = 0; for (int i = 0; i < (R < C ? R : C); i++) s += (*this)(i, i); return s; }
};

template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K> &x, const Mat<K, C> &y) {
  Mat<R, C> m;
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) {
      double s = 0;
      for (int k = 0; k < K; k++) s += x(r, k) * y(k, c);
      m(r, c) = s;
    }
  return m;
}
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C> &x) { return x * s; }

typedef Mat<3, 1> V3;
typedef Mat<3, 3> M3;
typedef Mat<6, 1> V6;
typedef Mat<6, 6> M6;

template <int N>
inline double dot(const Mat<N, 1> &x, const Mat<N, 1> &y) {
  double s = 0;
  for (int i = 0; i < N; i++) s += x[i] * y[i];
  return s;
}
inline V3 cross(const V3 &x, const V3 &y) {
  V3 c;
  c[0] = x[1] * y[2] - x[2] * y[1];
  c[1] = x[2] * y[0] - x[0] * y[2];
  c[2] = x[0] * y[1] - x[1] * y[0];
  return c;
}
inline V3 v3(double x, double y, double z) { V3 v; v[0] = x; v[1] = y; v[2] = z; return v; }

// Dynamic dense matrix / vector (row-major) for the LM normal equations
// (Eigen::MatrixXd / VectorXd in voxel_map.hpp:428-429, 630-631).
struct MatX {
  int rows = 0, cols = 0;
  std::vector<double> a;
  MatX() {}
  MatX(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  void resize(int r, int c) { rows = r; cols = c; a.assign((size_t)r * c, 0.0); }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  double &operator()(int r, int c) { return a[(size_t)r * cols + c]; }
  const double &operator()(int r, int c) const { return a[(size_t)r * cols + c]; }
  MatX &operator+=(const MatX &o) { for (size_t i = 0; i < a.size(); i++) a[i] += o.a[i]; return *this; }
  MatX &operator*=(double s) { for (size_t i = 0; i < a.size(); i++) a[i] *= s; return *this; }
  template <int BR, int BC>
  void addBlock(int r0, int c0, const Mat<BR, BC> &b) {
    for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) (*this)(r0 + r, c0 + c) += b(r, c);
  }
};
struct VecX {
  std::vector<double> a;
  VecX() {}
  explicit VecX(int n) : a(n, 0.0) {}
  void resize(int n) { a.assign(n, 0.0); }
  int size() const { return (int)a.size(); }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  double &operator[](int i) { return a[i]; }
  const double &operator[](int i) const { return a[i]; }
  VecX &operator+=(const VecX &o) { for (size_t i = 0; i < a.size(); i++) a[i] += o.a[i]; return *this; }
  VecX &operator*=(double s) { for (size_t i = 0; i < a.size(); i++) a[i] *= s; return *this; }
  template <int N>
  Mat<N, 1> seg(int i0) const { Mat<N, 1> v; for (int i = 0; i < N; i++) v[i] = a[i0 + i]; return v; }
  template <int N>
  void addSeg(int i0, const Mat<N, 1> &v) { for (int i = 0; i < N; i++) a[i0 + i] += v[i]; }
};

}  // namespace vso
