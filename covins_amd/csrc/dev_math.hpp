// dev_math.hpp — device-side geometry helpers for the gfx950 kernels (FP64 throughout: the reference's
// precision_t is double, covins_comm/include/covins/covins_base/typedefs_base.hpp:129, and IMU information
// matrices reach 1e14).
//
// Conventions: Hamilton quaternions stored [x,y,z,w]; pose = [q(4), p(3)] = T_w_s;
// pose tangent = [dtheta(3), dp(3)] with q+ = q (x) Exp(dtheta), p+ = p + dp.
// Everything is written on plain scalars / small fixed arrays so that it stays in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/covgpu.h"

#define COV_DEV __device__ __forceinline__

namespace covdev {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct M3 { double m[9]; };  // row-major

COV_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
COV_DEV V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
COV_DEV Q4 ldq(const double* p) { return Q4{p[0], p[1], p[2], p[3]}; }
COV_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
COV_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
COV_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
COV_DEV V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
COV_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
COV_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

COV_DEV Q4 qconj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
COV_DEV Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
COV_DEV Q4 qnormalize(Q4 q) {
  const double inv = 1.0 / sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Q4{q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}
COV_DEV M3 qrot(Q4 q) {
  M3 r;
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  r.m[0] = 1 - 2 * (yy + zz); r.m[1] = 2 * (xy - wz);     r.m[2] = 2 * (xz + wy);
  r.m[3] = 2 * (xy + wz);     r.m[4] = 1 - 2 * (xx + zz); r.m[5] = 2 * (yz - wx);
  r.m[6] = 2 * (xz - wy);     r.m[7] = 2 * (yz + wx);     r.m[8] = 1 - 2 * (xx + yy);
  return r;
}
COV_DEV V3 mul(const M3& a, V3 v) {
  return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
COV_DEV V3 mulT(const M3& a, V3 v) {  // a^T v
  return V3{a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
COV_DEV M3 mul(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
COV_DEV M3 transpose(const M3& a) {
  M3 t;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t.m[3 * i + j] = a.m[3 * j + i];
  return t;
}
COV_DEV M3 skew(V3 v) {
  M3 s;
  s.m[0] = 0;    s.m[1] = -v.z; s.m[2] = v.y;
  s.m[3] = v.z;  s.m[4] = 0;    s.m[5] = -v.x;
  s.m[6] = -v.y; s.m[7] = v.x;  s.m[8] = 0;
  return s;
}
COV_DEV M3 scaled(const M3& a, double s) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] * s;
  return c;
}
COV_DEV M3 add(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}
// e_w I + s [e_v]x   (s = +1: "L3", s = -1: "R3"; SURVEY.md A.1 helper)
COV_DEV M3 quat_lr3(Q4 e, double s) {
  M3 c;
  c.m[0] = e.w;        c.m[1] = -s * e.z;  c.m[2] = s * e.y;
  c.m[3] = s * e.z;    c.m[4] = e.w;       c.m[5] = -s * e.x;
  c.m[6] = -s * e.y;   c.m[7] = s * e.x;   c.m[8] = e.w;
  return c;
}
// Exp: rotation vector -> unit quaternion
COV_DEV Q4 qexp(V3 phi) {
  const double th2 = dot(phi, phi), th = sqrt(th2);
  const double s = (th < 1e-8) ? (0.5 - th2 / 48.0) : (sin(0.5 * th) / th);
  return Q4{s * phi.x, s * phi.y, s * phi.z, cos(0.5 * th)};
}

// ceres::CauchyLoss(a) with the Ceres 1.x corrector for rho'' < 0:
// returns sqrt(rho'), *cost = rho / 2; a <= 0 means "no loss".
COV_DEV double cauchy_scale(double a, double s, double* cost) {
  if (a <= 0.0) { *cost = 0.5 * s; return 1.0; }
  const double b = a * a, t = 1.0 + s / b;
  *cost = 0.5 * b * log(t);
  return sqrt(1.0 / t);
}

// clamp of Ceres' LM / dogleg diagonal: sqrt(h) in [1e-6, 1e32]
COV_DEV double clamp_diag(double h) { return fmin(fmax(sqrt(fmax(h, 0.0)), 1e-6), 1e32); }

// wave-level (64 lanes) sum, result valid in every lane
COV_DEV double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// R5: pinhole + radtan / equidistant. Returns false if the point is behind the camera (A.2: block zeroed).
COV_DEV bool project_point(V3 lc, const double* intr, const double* dist, int dist_type, double& u, double& v, double* jpi /*2x3*/) {
  if (!(lc.z > 1e-10)) return false;
  const double iz = 1.0 / lc.z, x = lc.x * iz, y = lc.y * iz;
  const double r2 = x * x + y * y;
  double xd, yd, dxx, dxy, dyx, dyy;
  if (dist_type == COVGPU_DIST_RADTAN) {
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const double rad = (k1 + k2 * r2) * r2, dr = k1 + 2.0 * k2 * r2;
    xd = x + x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    yd = y + y * rad + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
    const double xy2dr = 2.0 * x * y * dr;
    dxx = 1.0 + rad + 2.0 * x * x * dr + 2.0 * p1 * y + 6.0 * p2 * x;
    dxy = xy2dr + 2.0 * p1 * x + 2.0 * p2 * y;
    dyx = dxy;
    dyy = 1.0 + rad + 2.0 * y * y * dr + 2.0 * p2 * x + 6.0 * p1 * y;
  } else {
    const double rho = sqrt(r2);
    if (rho < 1e-8) {
      xd = x; yd = y; dxx = 1.0; dxy = 0.0; dyx = 0.0; dyy = 1.0;
    } else {
      const double th = atan(rho), t2 = th * th;
      const double poly = 1.0 + t2 * (dist[0] + t2 * (dist[1] + t2 * (dist[2] + t2 * dist[3])));
      const double dpoly = 1.0 + t2 * (3.0 * dist[0] + t2 * (5.0 * dist[1] + t2 * (7.0 * dist[2] + t2 * 9.0 * dist[3])));
      const double thd = th * poly, sc = thd / rho;
      const double dsc = (dpoly / (1.0 + r2) * rho - thd) / r2;
      const double ir = 1.0 / rho;
      xd = sc * x; yd = sc * y;
      dxx = sc + x * x * dsc * ir; dxy = x * y * dsc * ir; dyx = dxy; dyy = sc + y * y * dsc * ir;
    }
  }
  u = intr[0] * xd + intr[2];
  v = intr[1] * yd + intr[3];
  if (jpi) {
    jpi[0] = intr[0] * dxx * iz; jpi[1] = intr[0] * dxy * iz; jpi[2] = -intr[0] * (dxx * x + dxy * y) * iz;
    jpi[3] = intr[1] * dyx * iz; jpi[4] = intr[1] * dyy * iz; jpi[5] = -intr[1] * (dyx * x + dyy * y) * iz;
  }
  return true;
}

}  // namespace covdev
