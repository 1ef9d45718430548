// inertial_dev.hpp — device functions of the IMU preintegration factor (R3, SURVEY.md A.4: robopt::imu::PreintegrationFactor,
// optimization_be.cpp:415-416) shared by k_inertial.hip and k_tail.hip (the fused trust-region tail). One WAVE per factor,
// kImuWaves factors per workgroup, working matrices in the wave's LDS slice.
#pragma once
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

COV_DEV void set3(double* M, int ldm, int r0, int c0, const M3& b, double s) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[(r0 + r) * ldm + c0 + c] = b.m[3 * r + c] * s;
}
COV_DEV M3 ident3() { M3 i; for (int k = 0; k < 9; ++k) i.m[k] = 0; i.m[0] = i.m[4] = i.m[8] = 1; return i; }
COV_DEV M3 get3(const double* M, int ldm, int r0, int c0) {
  M3 b;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) b.m[3 * r + c] = M[(r0 + r) * ldm + c0 + c];
  return b;
}

constexpr int kImuWaves = 4;  // factors per workgroup

// R3, un-whitened: residual u[15] and (optionally) the 15x30 Jacobian A, parameter order
// [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]. A must be zero-filled by the caller. Runs on ONE lane.
template <bool JAC>
COV_DEV void imu_unwhitened(const DevProblem& P, const double* __restrict__ pose, const double* __restrict__ sb, int f, double* u, double* A) {
  const int i = P.imu_i[f], j = P.imu_j[f];
  const double* d = P.pre_delta + 11 * (size_t)f;
  const double* PJ = P.pre_J + 225 * (size_t)f;
  const double* pb = P.pre_bias + 6 * (size_t)f;
  const double *Ti = pose + 7 * i, *Tj = pose + 7 * j, *si = sb + 9 * i, *sj = sb + 9 * j;
  const Q4 qi = ldq(Ti), qj = ldq(Tj);
  const V3 pi = ld3(Ti + 4), pj = ld3(Tj + 4);
  const V3 vi = ld3(si), bai = ld3(si + 3), bgi = ld3(si + 6);
  const V3 vj = ld3(sj), baj = ld3(sj + 3), bgj = ld3(sj + 6);
  const V3 dba = bai - ld3(pb), dbg = bgi - ld3(pb + 3);
  const M3 Jp_ba = get3(PJ, 15, 0, 9), Jp_bg = get3(PJ, 15, 0, 12), Jq_bg = get3(PJ, 15, 3, 12);
  const M3 Jv_ba = get3(PJ, 15, 6, 9), Jv_bg = get3(PJ, 15, 6, 12);
  const V3 hq = mul(Jq_bg, dbg) * 0.5;
  const Q4 dq = ldq(d + 3);
  const Q4 dqc = qnormalize(qmul(dq, Q4{hq.x, hq.y, hq.z, 1.0}));
  const V3 dvc = ld3(d + 7) + mul(Jv_ba, dba) + mul(Jv_bg, dbg);
  const V3 dpc = ld3(d) + mul(Jp_ba, dba) + mul(Jp_bg, dbg);
  const double dt = d[10], g = P.imu_noise[5 * (size_t)f + 4];
  const M3 Ri = qrot(qi);
  const V3 tp = mulT(Ri, V3{pj.x - pi.x - vi.x * dt, pj.y - pi.y - vi.y * dt, pj.z - pi.z - vi.z * dt + 0.5 * g * dt * dt});
  const V3 tv = mulT(Ri, V3{vj.x - vi.x, vj.y - vi.y, vj.z - vi.z + g * dt});
  const Q4 qij = qmul(qconj(qi), qj);
  const Q4 e = qmul(qconj(dqc), qij);
  u[0] = tp.x - dpc.x; u[1] = tp.y - dpc.y; u[2] = tp.z - dpc.z;
  u[3] = 2.0 * e.x; u[4] = 2.0 * e.y; u[5] = 2.0 * e.z;
  u[6] = tv.x - dvc.x; u[7] = tv.y - dvc.y; u[8] = tv.z - dvc.z;
  u[9] = baj.x - bai.x; u[10] = baj.y - bai.y; u[11] = baj.z - bai.z;
  u[12] = bgj.x - bgi.x; u[13] = bgj.y - bgi.y; u[14] = bgj.z - bgi.z;
  if (!JAC) return;
  const M3 RiT = transpose(Ri), I3 = ident3();
  const bool fi = P.fixed[i] != 0, fj = P.fixed[j] != 0;
  if (!fi) {
    set3(A, 30, 0, 0, skew(tp), 1.0);
    set3(A, 30, 0, 3, RiT, -1.0);
    // -(Lq(a) Rq(b))_vv, a = q_j^-1 q_i, b = dq_c :  L3(a) R3(b) - a_v b_v^T
    const Q4 a = qconj(qij);
    M3 lr = mul(quat_lr3(a, 1.0), quat_lr3(dqc, -1.0));
    const double av[3] = {a.x, a.y, a.z}, bv[3] = {dqc.x, dqc.y, dqc.z};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c) lr.m[3 * rr + c] -= av[rr] * bv[c];
    set3(A, 30, 3, 0, lr, -1.0);
    set3(A, 30, 6, 0, skew(tv), 1.0);
  }
  set3(A, 30, 0, 6, RiT, -dt);
  set3(A, 30, 0, 9, Jp_ba, -1.0);
  set3(A, 30, 0, 12, Jp_bg, -1.0);
  set3(A, 30, 3, 12, mul(quat_lr3(qconj(e), 1.0), Jq_bg), -1.0);
  set3(A, 30, 6, 6, RiT, -1.0);
  set3(A, 30, 6, 9, Jv_ba, -1.0);
  set3(A, 30, 6, 12, Jv_bg, -1.0);
  set3(A, 30, 9, 9, I3, -1.0);
  set3(A, 30, 12, 12, I3, -1.0);
  if (!fj) {
    set3(A, 30, 0, 18, RiT, 1.0);
    set3(A, 30, 3, 15, quat_lr3(e, 1.0), 1.0);
  }
  set3(A, 30, 6, 21, RiT, 1.0);
  set3(A, 30, 9, 24, I3, 1.0);
  set3(A, 30, 12, 27, I3, 1.0);
}

// Per-wave LDS layout of one factor: A[450] | J[450] | W[225] | u[15] | r[15]
constexpr int kImuLds = 450 + 450 + 225 + 16 + 16;

// whitened residual r and (JAC) whitened Jacobian J of factor f in this wave's LDS slice; contains workgroup barriers
template <bool JAC>
COV_DEV void imu_stage(const DevProblem& P, const double* __restrict__ pose, const double* __restrict__ sb, int f, bool live, int lane,
                       double* sl) {
  double* A = sl; double* J = sl + 450; double* W = sl + 900; double* u = sl + 1125; double* r = sl + 1141;
  if (JAC) for (int k = lane; k < 450; k += 64) A[k] = 0.0;
  if (live) for (int k = lane; k < 225; k += 64) W[k] = P.pre_W[225 * (size_t)f + k];
  __syncthreads();
  if (live && lane == 0) imu_unwhitened<JAC>(P, pose, sb, f, u, A);
  __syncthreads();
  if (live) {
    if (lane < 15) {
      double s2 = 0.0;
      for (int k = 0; k <= lane; ++k) s2 += W[15 * lane + k] * u[k];
      r[lane] = s2;
    }
    if (JAC)
      for (int e = lane; e < 450; e += 64) {
        const int rr = e / 30, c = e - 30 * rr;
        double s2 = 0.0;
        for (int k = 0; k <= rr; ++k) s2 += W[15 * rr + k] * A[30 * k + c];
        J[e] = s2;
      }
  }
  __syncthreads();
}

}  // namespace covgpu
