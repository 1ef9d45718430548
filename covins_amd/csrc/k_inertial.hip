// k_inertial.hip — IMU preintegration and the preintegration factor on gfx950.
//
// Replaces (sources un-vendored, robopt_open@fix_imu_residual):
//   R2 robopt::imu::PreintegrationBase::repropagate(ba, bg)   called once per keyframe per solve at
//      covins_backend/src/covins_backend/optimization_be.cpp:396 (API also at keyframe_be.cpp:187-203)
//   R3 robopt::imu::PreintegrationFactor  SizedCostFunction<15,7,9,7,9>, no loss (opt_be.cpp:415-416)
//
// Round-1 shape: one thread per factor with the 15x15 working matrices in per-lane scratch. There are only
// ~K factors (2k on the 5-agent map) and preintegration runs once per solve, so these kernels are latency-,
// not throughput-relevant; DESIGN.md §6 lists the one-wave-per-factor LDS version as the planned upgrade.
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

// R2. Midpoint scheme; state order [P, R, V, BA, BG]; noise order [n_a0, n_g0, n_a1, n_g1, n_ba, n_bg].
__global__ __launch_bounds__(64) void k_preintegrate(DevProblem P, double sa, double sg, double saw, double sgw) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= P.I) return;
  const int j = P.imu_j[f];
  const V3 ba = ld3(P.sb + 9 * j + 3), bg = ld3(P.sb + 9 * j + 6);
  double J[225], C[225], F[225], T[225], V[270];
  for (int k = 0; k < 225; ++k) { J[k] = 0.0; C[k] = 0.0; }
  for (int k = 0; k < 15; ++k) J[16 * k] = 1.0;
  V3 dp = v3(0, 0, 0), dv = v3(0, 0, 0);
  Q4 dq = Q4{0, 0, 0, 1};
  double dtsum = 0.0;
  V3 a0 = ld3(P.imu_first + 6 * f), w0 = ld3(P.imu_first + 6 * f + 3);
  const double nd[6] = {sa * sa, sg * sg, sa * sa, sg * sg, saw * saw, sgw * sgw};
  const M3 I3 = ident3();
  for (int s = P.imu_ptr[f]; s < P.imu_ptr[f + 1]; ++s) {
    const double* sm = P.imu_samples + 7 * (size_t)s;
    const double dt = sm[0];
    const V3 a1 = ld3(sm + 1), w1 = ld3(sm + 4);
    const V3 w = (w0 + w1) * 0.5 - bg;
    const Q4 dq1 = qnormalize(qmul(dq, Q4{w.x * dt * 0.5, w.y * dt * 0.5, w.z * dt * 0.5, 1.0}));
    const M3 Rq = qrot(dq), Rr = qrot(dq1);
    const V3 abar = (mul(Rq, a0 - ba) + mul(Rr, a1 - ba)) * 0.5;
    const V3 dp1 = dp + dv * dt + abar * (0.5 * dt * dt);
    const V3 dv1 = dv + abar * dt;
    const M3 RqA0 = mul(Rq, skew(a0 - ba)), RrA1 = mul(Rr, skew(a1 - ba));
    const M3 ImO = add(I3, scaled(skew(w), -dt));
    const M3 RrA1ImO = mul(RrA1, ImO);
    const M3 RqRr = add(Rq, Rr);
    for (int k = 0; k < 225; ++k) F[k] = 0.0;
    for (int k = 0; k < 270; ++k) V[k] = 0.0;
    const double dt2 = dt * dt, dt3 = dt2 * dt;
    set3(F, 15, 0, 0, I3, 1.0);
    set3(F, 15, 0, 3, add(scaled(RqA0, -0.25 * dt2), scaled(RrA1ImO, -0.25 * dt2)), 1.0);
    set3(F, 15, 0, 6, I3, dt);
    set3(F, 15, 0, 9, RqRr, -0.25 * dt2);
    set3(F, 15, 0, 12, RrA1, 0.25 * dt3);
    set3(F, 15, 3, 3, ImO, 1.0);
    set3(F, 15, 3, 12, I3, -dt);
    set3(F, 15, 6, 3, add(scaled(RqA0, -0.5 * dt), scaled(RrA1ImO, -0.5 * dt)), 1.0);
    set3(F, 15, 6, 6, I3, 1.0);
    set3(F, 15, 6, 9, RqRr, -0.5 * dt);
    set3(F, 15, 6, 12, RrA1, 0.5 * dt2);
    set3(F, 15, 9, 9, I3, 1.0);
    set3(F, 15, 12, 12, I3, 1.0);
    set3(V, 18, 0, 0, Rq, 0.25 * dt2);
    set3(V, 18, 0, 3, RrA1, -0.125 * dt3);
    set3(V, 18, 0, 6, Rr, 0.25 * dt2);
    set3(V, 18, 0, 9, RrA1, -0.125 * dt3);
    set3(V, 18, 3, 3, I3, 0.5 * dt);
    set3(V, 18, 3, 9, I3, 0.5 * dt);
    set3(V, 18, 6, 0, Rq, 0.5 * dt);
    set3(V, 18, 6, 3, RrA1, -0.25 * dt2);
    set3(V, 18, 6, 6, Rr, 0.5 * dt);
    set3(V, 18, 6, 9, RrA1, -0.25 * dt2);
    set3(V, 18, 9, 12, I3, dt);
    set3(V, 18, 12, 15, I3, dt);
    // J <- F J
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) {
        double s2 = 0.0;
        for (int k = 0; k < 15; ++k) s2 += F[15 * r + k] * J[15 * k + c];
        T[15 * r + c] = s2;
      }
    for (int k = 0; k < 225; ++k) J[k] = T[k];
    // C <- F C F^T + V N V^T
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) {
        double s2 = 0.0;
        for (int k = 0; k < 15; ++k) s2 += F[15 * r + k] * C[15 * k + c];
        T[15 * r + c] = s2;
      }
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) {
        double s2 = 0.0;
        for (int k = 0; k < 15; ++k) s2 += T[15 * r + k] * F[15 * c + k];
        for (int k = 0; k < 18; ++k) s2 += V[18 * r + k] * nd[k / 3] * V[18 * c + k];
        C[15 * r + c] = s2;
      }
    dp = dp1; dv = dv1; dq = dq1; dtsum += dt;
    a0 = a1; w0 = w1;
  }
  double* d = P.pre_delta + 11 * (size_t)f;
  d[0] = dp.x; d[1] = dp.y; d[2] = dp.z; d[3] = dq.x; d[4] = dq.y; d[5] = dq.z; d[6] = dq.w;
  d[7] = dv.x; d[8] = dv.y; d[9] = dv.z; d[10] = dtsum;
  double* pb = P.pre_bias + 6 * (size_t)f;
  pb[0] = ba.x; pb[1] = ba.y; pb[2] = ba.z; pb[3] = bg.x; pb[4] = bg.y; pb[5] = bg.z;
  for (int k = 0; k < 225; ++k) { P.pre_J[225 * (size_t)f + k] = J[k]; P.pre_P[225 * (size_t)f + k] = C[k]; }
  // whitening W = chol(C)^-1 (lower): ||W r||^2 = r^T C^-1 r without ever forming C^-1 (cond(C) ~ 2e8, A.4)
  bool ok = true;
  for (int c = 0; c < 15 && ok; ++c) {
    double dd = C[16 * c];
    for (int k = 0; k < c; ++k) dd -= C[15 * c + k] * C[15 * c + k];
    if (!(dd > 0.0)) { ok = false; break; }
    dd = sqrt(dd);
    C[16 * c] = dd;
    for (int r = c + 1; r < 15; ++r) {
      double s2 = C[15 * r + c];
      for (int k = 0; k < c; ++k) s2 -= C[15 * r + k] * C[15 * c + k];
      C[15 * r + c] = s2 / dd;
    }
  }
  double* Wm = P.pre_W + 225 * (size_t)f;
  for (int k = 0; k < 225; ++k) Wm[k] = 0.0;
  if (ok) {
    for (int c = 0; c < 15; ++c) {
      T[16 * c] = 1.0 / C[16 * c];
      for (int r = c + 1; r < 15; ++r) {
        double s2 = 0.0;
        for (int k = c; k < r; ++k) s2 += C[15 * r + k] * T[15 * k + c];
        T[15 * r + c] = -s2 / C[16 * r];
      }
    }
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c <= r; ++c) Wm[15 * r + c] = T[15 * r + c];
  }
}

// R3: whitened residual r[15] and (optionally) whitened Jacobian Jw[15x30], parameter order
// [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]. `A` is caller-provided 15x30 scratch.
template <bool JAC>
COV_DEV void eval_imu(const DevProblem& P, const double* __restrict__ pose, const double* __restrict__ sb, int f, double* r, double* Jw,
                      double* A) {
  const int i = P.imu_i[f], j = P.imu_j[f];
  const double* d = P.pre_delta + 11 * (size_t)f;
  const double* PJ = P.pre_J + 225 * (size_t)f;
  const double* Wm = P.pre_W + 225 * (size_t)f;
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
  const double dt = d[10], g = P.gravity;
  const M3 Ri = qrot(qi);
  const V3 tp = mulT(Ri, V3{pj.x - pi.x - vi.x * dt, pj.y - pi.y - vi.y * dt, pj.z - pi.z - vi.z * dt + 0.5 * g * dt * dt});
  const V3 tv = mulT(Ri, V3{vj.x - vi.x, vj.y - vi.y, vj.z - vi.z + g * dt});
  const Q4 qij = qmul(qconj(qi), qj);
  const Q4 e = qmul(qconj(dqc), qij);
  double u[15];
  u[0] = tp.x - dpc.x; u[1] = tp.y - dpc.y; u[2] = tp.z - dpc.z;
  u[3] = 2.0 * e.x; u[4] = 2.0 * e.y; u[5] = 2.0 * e.z;
  u[6] = tv.x - dvc.x; u[7] = tv.y - dvc.y; u[8] = tv.z - dvc.z;
  u[9] = baj.x - bai.x; u[10] = baj.y - bai.y; u[11] = baj.z - bai.z;
  u[12] = bgj.x - bgi.x; u[13] = bgj.y - bgi.y; u[14] = bgj.z - bgi.z;
  for (int rr = 0; rr < 15; ++rr) {
    double s2 = 0.0;
    for (int k = 0; k <= rr; ++k) s2 += Wm[15 * rr + k] * u[k];
    r[rr] = s2;
  }
  if (!JAC) return;
  for (int k = 0; k < 450; ++k) A[k] = 0.0;
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
  for (int rr = 0; rr < 15; ++rr)
    for (int c = 0; c < 30; ++c) {
      double s2 = 0.0;
      for (int k = 0; k <= rr; ++k) s2 += Wm[15 * rr + k] * A[30 * k + c];
      Jw[30 * rr + c] = s2;
    }
}

__global__ __launch_bounds__(64) void k_imu_build(DevProblem P) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= P.I) return;
  double r[15], Jw[450], A[450];
  eval_imu<true>(P, P.pose, P.sb, f, r, Jw, A);
  const int i = P.imu_i[f], j = P.imu_j[f];
  const size_t ld = (size_t)P.npad;
  double cost = 0.0;
  for (int k = 0; k < 15; ++k) cost += r[k] * r[k];
  atomicAdd(&P.scal[SC_COST], 0.5 * cost);
  // Scatter J^T J (30x30; groups P_i(0:6) S_i(6:15) P_j(15:21) S_j(21:30)) into the structured system:
  //   pose-pose  -> C (dense, chain-major order);  sb-sb -> Ad / Ae;  sb-pose -> Bs / Bn / Bp   
  // By construction of perm[], j sits right after i on the same chain: pj == pi + 1.
  const int pi = P.perm[i], pj = P.perm[j];
  for (int a = 0; a < 30; ++a) {
    const int ga_ = a / 15, la = a % 15;                 // keyframe (0 = i, 1 = j), local dim
    const int ra = 15 * (ga_ ? j : i) + la;
    double ga = 0.0;
    for (int k = 0; k < 15; ++k) ga += Jw[30 * k + a] * r[k];
    if (ga != 0.0) { atomicAdd(P.grad + ra, ga); atomicAdd(P.bred + ra, -ga); }
    for (int b = 0; b < 30; ++b) {
      const int gb_ = b / 15, lb = b % 15;
      double h = 0.0;
      for (int k = 0; k < 15; ++k) h += Jw[30 * k + a] * Jw[30 * k + b];
      if (h == 0.0) continue;
      if (a == b) atomicAdd(P.hdiag + ra, h);
      const int posa = ga_ ? pj : pi, posb = gb_ ? pj : pi;
      if (la < 6 && lb < 6) {          // pose-pose: lower triangle of C
        const int ca = 6 * posa + la, cb = 6 * posb + lb;
        if (cb <= ca) atomicAdd(P.Sred + (size_t)ca * ld + cb, h);
      } else if (la >= 6 && lb >= 6) { // sb-sb: full diagonal blocks, sub-diagonal block (pos_j, pos_i)
        if (ga_ == gb_) atomicAdd(P.Ad + (size_t)81 * posa + 9 * (la - 6) + (lb - 6), h);
        else if (ga_ == 1) atomicAdd(P.Ae + (size_t)81 * pj + 9 * (la - 6) + (lb - 6), h);
      } else if (la >= 6) {            // sb (row) x pose (col)
        double* blk = (ga_ == gb_) ? P.Bs : (ga_ == 0 ? P.Bn : P.Bp);   // same kf | sb_i x pose_j (next) | sb_j x pose_i (prev)
        atomicAdd(blk + (size_t)54 * posa + 6 * (la - 6) + lb, h);
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_imu_jvp(DevProblem P, const double* __restrict__ v_all) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (f < P.I) {
    double r[15], Jw[450], A[450];
    eval_imu<true>(P, P.pose, P.sb, f, r, Jw, A);
    const double* vi = v_all + 15 * (size_t)P.imu_i[f];
    const double* vj = v_all + 15 * (size_t)P.imu_j[f];
    for (int k = 0; k < 15; ++k) {
      double s2 = 0.0;
      for (int c = 0; c < 15; ++c) s2 += Jw[30 * k + c] * vi[c] + Jw[30 * k + 15 + c] * vj[c];
      acc += s2 * s2;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(&P.scal[SC_JV2], acc);
}

__global__ __launch_bounds__(64) void k_imu_cost(DevProblem P, const double* __restrict__ pose, const double* __restrict__ sb) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (f < P.I) {
    double r[15];
    eval_imu<false>(P, pose, sb, f, r, nullptr, nullptr);
    for (int k = 0; k < 15; ++k) acc += r[k] * r[k];
  }
  acc = wave_sum(0.5 * acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(&P.scal[SC_COST], acc);
}

__global__ __launch_bounds__(64) void k_imu_linearize(DevProblem P, double* r_out, double* J_out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= P.I) return;
  double r[15], Jw[450], A[450];
  eval_imu<true>(P, P.pose, P.sb, f, r, Jw, A);
  for (int k = 0; k < 15; ++k) r_out[15 * (size_t)f + k] = r[k];
  for (int k = 0; k < 450; ++k) J_out[450 * (size_t)f + k] = Jw[k];
}

void launch_preintegrate(const DevProblem& P, const covgpu_options& o, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_preintegrate, dim3((P.I + 63) / 64), dim3(64), 0, st, P, o.sigma_a, o.sigma_g, o.sigma_aw, o.sigma_gw);
}
void launch_imu_build(const DevProblem& P, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_build, dim3((P.I + 63) / 64), dim3(64), 0, st, P);
}
void launch_imu_jvp(const DevProblem& P, const double* v_all, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_jvp, dim3((P.I + 63) / 64), dim3(64), 0, st, P, v_all);
}
void launch_imu_cost(const DevProblem& P, const double* pose, const double* sb, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_cost, dim3((P.I + 63) / 64), dim3(64), 0, st, P, pose, sb);
}
void launch_imu_linearize(const DevProblem& P, double* r, double* J, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_linearize, dim3((P.I + 63) / 64), dim3(64), 0, st, P, r, J);
}

}  // namespace covgpu
