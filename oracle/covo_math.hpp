// covo_math.hpp — tiny fixed-size linear algebra for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing under oracle/ is linked into, imported by
// or called from the product path (covins_amd/, include/). It exists so that tests/ and bench.py's
// cpu_baseline leg can check / time the HIP path against a plain CPU restatement.
//
// No Eigen exists in this image (container facts), hence this header.
#pragma once
#include <cmath>
#include <cstring>
#include <initializer_list>

namespace covo {

template <int R, int C>
struct Mat {
  double a[R * C];
  Mat() { std::memset(a, 0, sizeof(a)); }
  double& operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  double& operator[](int i) { return a[i]; }
  double operator[](int i) const { return a[i]; }
  static Mat Identity() {
    Mat m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
  Mat<C, R> T() const {
    Mat<C, R> t;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c);
    return t;
  }
  Mat operator+(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] + o.a[i]; return m; }
  Mat operator-(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] - o.a[i]; return m; }
  Mat operator-() const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = -a[i]; return m; }
  Mat operator*(double s) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] * s; return m; }
  Mat& operator+=(const Mat& o) { for (int i = 0; i < R * C; ++i) a[i] += o.a[i]; return *this; }
  Mat& operator-=(const Mat& o) { for (int i = 0; i < R * C; ++i) a[i] -= o.a[i]; return *this; }
  template <int BR, int BC>
  Mat<BR, BC> block(int r0, int c0) const {
    Mat<BR, BC> b;
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) b(r, c) = (*this)(r0 + r, c0 + c);
    return b;
  }
  template <int BR, int BC>
  void set(int r0, int c0, const Mat<BR, BC>& b) {
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
  }
  double squaredNorm() const { double s = 0; for (int i = 0; i < R * C; ++i) s += a[i] * a[i]; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
};

template <int R, int K, int C>
Mat<R, C> operator*(const Mat<R, K>& A, const Mat<K, C>& B) {
  Mat<R, C> m;
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) {
      const double v = A(r, k);
      if (v == 0.0) continue;
      for (int c = 0; c < C; ++c) m(r, c) += v * B(k, c);
    }
  return m;
}
template <int R, int C>
Mat<R, C> operator*(double s, const Mat<R, C>& A) { return A * s; }

using Vec3 = Mat<3, 1>;
using Mat3 = Mat<3, 3>;
using Vec6 = Mat<6, 1>;

inline Vec3 v3(double x, double y, double z) { Vec3 v; v[0] = x; v[1] = y; v[2] = z; return v; }
inline Vec3 v3(const double* p) { return v3(p[0], p[1], p[2]); }
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline Mat3 skew(const Vec3& v) {
  Mat3 m;
  m(0, 1) = -v[2]; m(0, 2) = v[1];
  m(1, 0) = v[2];  m(1, 2) = -v[0];
  m(2, 0) = -v[1]; m(2, 1) = v[0];
  return m;
}

// Hamilton unit quaternion stored [x,y,z,w] (keyframe_base.cpp:493-499).
struct Quat {
  double x = 0, y = 0, z = 0, w = 1;
  Quat() {}
  Quat(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
  explicit Quat(const double* p) : x(p[0]), y(p[1]), z(p[2]), w(p[3]) {}
  Vec3 vec() const { return v3(x, y, z); }
  Quat inv() const { return Quat(-x, -y, -z, w); }
  Quat normalized() const {
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    return Quat(x / n, y / n, z / n, w / n);
  }
  Quat operator*(const Quat& b) const {
    return Quat(w * b.x + x * b.w + y * b.z - z * b.y,
                w * b.y - x * b.z + y * b.w + z * b.x,
                w * b.z + x * b.y - y * b.x + z * b.w,
                w * b.w - x * b.x - y * b.y - z * b.z);
  }
  Mat3 R() const {
    Mat3 m;
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    m(0, 0) = 1 - 2 * (yy + zz); m(0, 1) = 2 * (xy - wz);     m(0, 2) = 2 * (xz + wy);
    m(1, 0) = 2 * (xy + wz);     m(1, 1) = 1 - 2 * (xx + zz); m(1, 2) = 2 * (yz - wx);
    m(2, 0) = 2 * (xz - wy);     m(2, 1) = 2 * (yz + wx);     m(2, 2) = 1 - 2 * (xx + yy);
    return m;
  }
};

// Exp: rotation vector -> unit quaternion
inline Quat quat_exp(const Vec3& phi) {
  const double th = phi.norm();
  double s;  // sin(th/2)/th
  if (th < 1e-8) s = 0.5 - th * th / 48.0; else s = std::sin(0.5 * th) / th;
  return Quat(s * phi[0], s * phi[1], s * phi[2], std::cos(0.5 * th));
}
// bottom-right 3x3 of left / right quaternion product matrices (SURVEY.md A.1 helper)
inline Mat3 quat_L3(const Quat& e) { return Mat3::Identity() * e.w + skew(e.vec()); }
inline Mat3 quat_R3(const Quat& e) { return Mat3::Identity() * e.w - skew(e.vec()); }

// rotation matrix -> quaternion (Shepperd), w >= 0
inline Quat quat_from_R(const Mat3& m) {
  Quat q;
  const double tr = m(0, 0) + m(1, 1) + m(2, 2);
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    q.w = 0.25 * s; q.x = (m(2, 1) - m(1, 2)) / s; q.y = (m(0, 2) - m(2, 0)) / s; q.z = (m(1, 0) - m(0, 1)) / s;
  } else if (m(0, 0) > m(1, 1) && m(0, 0) > m(2, 2)) {
    double s = std::sqrt(1.0 + m(0, 0) - m(1, 1) - m(2, 2)) * 2;
    q.w = (m(2, 1) - m(1, 2)) / s; q.x = 0.25 * s; q.y = (m(0, 1) + m(1, 0)) / s; q.z = (m(0, 2) + m(2, 0)) / s;
  } else if (m(1, 1) > m(2, 2)) {
    double s = std::sqrt(1.0 + m(1, 1) - m(0, 0) - m(2, 2)) * 2;
    q.w = (m(0, 2) - m(2, 0)) / s; q.x = (m(0, 1) + m(1, 0)) / s; q.y = 0.25 * s; q.z = (m(1, 2) + m(2, 1)) / s;
  } else {
    double s = std::sqrt(1.0 + m(2, 2) - m(0, 0) - m(1, 1)) * 2;
    q.w = (m(1, 0) - m(0, 1)) / s; q.x = (m(0, 2) + m(2, 0)) / s; q.y = (m(1, 2) + m(2, 1)) / s; q.z = 0.25 * s;
  }
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  return q.normalized();
}

// in-place lower Cholesky of an NxN SPD matrix; returns false if a pivot is <= 0
template <int N>
bool chol_lower(Mat<N, N>& A) {
  for (int j = 0; j < N; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < N; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
      A(i, j) = s / d;
    }
    for (int c = j + 1; c < N; ++c) A(j, c) = 0.0;
  }
  return true;
}
// inverse of a lower-triangular matrix
template <int N>
Mat<N, N> tri_lower_inverse(const Mat<N, N>& L) {
  Mat<N, N> X;
  for (int c = 0; c < N; ++c) {
    X(c, c) = 1.0 / L(c, c);
    for (int r = c + 1; r < N; ++r) {
      double s = 0;
      for (int k = c; k < r; ++k) s += L(r, k) * X(k, c);
      X(r, c) = -s / L(r, r);
    }
  }
  return X;
}
// symmetric 3x3 inverse via adjugate
inline bool inv3_sym(const Mat3& A, Mat3& X) {
  const double a = A(0, 0), b = A(0, 1), c = A(0, 2), d = A(1, 1), e = A(1, 2), f = A(2, 2);
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(std::fabs(det) > 0.0)) return false;
  const double id = 1.0 / det;
  X(0, 0) = c00 * id; X(0, 1) = c01 * id; X(0, 2) = c02 * id;
  X(1, 0) = X(0, 1);  X(1, 1) = (a * f - c * c) * id; X(1, 2) = (b * c - a * e) * id;
  X(2, 0) = X(0, 2);  X(2, 1) = X(1, 2); X(2, 2) = (a * d - b * b) * id;
  return true;
}

}  // namespace covo
