// k_inertial.hip — IMU preintegration and the preintegration factor on gfx950.
//
// Replaces (sources un-vendored, robopt_open@fix_imu_residual):
//   R2 robopt::imu::PreintegrationBase::repropagate(ba, bg)   called once per keyframe per solve at
//      covins_backend/src/covins_backend/optimization_be.cpp:396 (API also at keyframe_be.cpp:187-203)
//   R3 robopt::imu::PreintegrationFactor  SizedCostFunction<15,7,9,7,9>, no loss (opt_be.cpp:415-416)
//
// One WAVE per factor, four factors per workgroup, all 15x15 / 15x30 working matrices in LDS (~9 KB per factor):
// lane 0 does the short geometric part (quaternions, 3x3 blocks), the 64 lanes share the matrix products entry by
// entry. The first version (one thread per factor, matrices in per-lane scratch) took 14.6 ms for preintegration
// and 1.2 ms per build on the 5-agent map; these take ~0.3 ms and ~0.05 ms.
#include "common.hpp"
#include "dev_math.hpp"
#include "reduce.hpp"
#include "inertial_dev.hpp"

namespace covgpu {
using namespace covdev;

// R2. Midpoint scheme; state order [P, R, V, BA, BG]; noise order [n_a0, n_g0, n_a1, n_g1, n_ba, n_bg].
__global__ __launch_bounds__(64 * kImuWaves) void k_preintegrate(DevProblem P) {
  __shared__ double sm[kImuWaves][225 * 4 + 270];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * kImuWaves + wave;
  const bool live = f < P.I;
  double* F = sm[wave];
  double* J = F + 225;
  double* C = J + 225;
  double* T = C + 225;
  double* V = T + 225;
  const int fj = live ? P.imu_j[f] : 0;
  const V3 ba = ld3(P.sb + 9 * fj + 3), bg = ld3(P.sb + 9 * fj + 6);
  for (int k = lane; k < 225; k += 64) { F[k] = 0.0; J[k] = (k % 16 == 0) ? 1.0 : 0.0; C[k] = 0.0; }
  for (int k = lane; k < 270; k += 64) V[k] = 0.0;
  V3 dp = v3(0, 0, 0), dv = v3(0, 0, 0);
  Q4 dq = Q4{0, 0, 0, 1};
  double dtsum = 0.0;
  V3 a0 = live ? ld3(P.imu_first + 6 * (size_t)f) : v3(0, 0, 0), w0 = live ? ld3(P.imu_first + 6 * (size_t)f + 3) : v3(0, 0, 0);
  const double* nz = P.imu_noise + 5 * (size_t)(live ? f : 0);  // this factor's own calibration (keyframe_be.cpp:187-195)
  const double sa = nz[0], sg = nz[1], saw = nz[2], sgw = nz[3];
  const double nd[6] = {sa * sa, sg * sg, sa * sa, sg * sg, saw * saw, sgw * sgw};
  const M3 I3 = ident3();
  // every wave of the workgroup runs the same number of barrier-carrying steps
  __shared__ int s_steps[kImuWaves];
  if (lane == 0) s_steps[wave] = live ? P.imu_ptr[f + 1] - P.imu_ptr[f] : 0;
  __syncthreads();
  int nmax = 0;
#pragma unroll
  for (int k = 0; k < kImuWaves; ++k) nmax = max(nmax, s_steps[k]);
  const int s0 = live ? P.imu_ptr[f] : 0, ns = s_steps[wave];
  for (int it = 0; it < nmax; ++it) {
    const bool on = it < ns;
    if (on && lane == 0) {  // geometric part: blocks of F and V (constant identity blocks are rewritten too: cheap)
      const double* smp = P.imu_samples + 7 * (size_t)(s0 + it);
      const double dt = smp[0];
      const V3 a1 = ld3(smp + 1), w1 = ld3(smp + 4);
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
      dp = dp1; dv = dv1; dq = dq1; dtsum += dt;
      a0 = a1; w0 = w1;
    }
    __syncthreads();
    // J <- F J   (entries in registers first: in-place needs the old J)
    double jn[4], tn[4];
    if (on) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        if (e < 225) {
          const int r = e / 15, c = e - 15 * r;
          double s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) { s2 += F[15 * r + k] * J[15 * k + c]; s3 += F[15 * r + k] * C[15 * k + c]; }
          jn[u] = s2; tn[u] = s3;
        }
      }
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        if (e < 225) { J[e] = jn[u]; T[e] = tn[u]; }
      }
    }
    __syncthreads();
    // C <- (F C) F^T + V N V^T
    if (on) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        if (e < 225) {
          const int r = e / 15, c = e - 15 * r;
          double s2 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) s2 += T[15 * r + k] * F[15 * c + k];
#pragma unroll
          for (int k = 0; k < 18; ++k) s2 += V[18 * r + k] * nd[k / 3] * V[18 * c + k];
          C[e] = s2;
        }
      }
    }
    __syncthreads();
  }
  if (!live) return;  // no barriers below
  if (lane == 0) {
    double* d = P.pre_delta + 11 * (size_t)f;
    d[0] = dp.x; d[1] = dp.y; d[2] = dp.z; d[3] = dq.x; d[4] = dq.y; d[5] = dq.z; d[6] = dq.w;
    d[7] = dv.x; d[8] = dv.y; d[9] = dv.z; d[10] = dtsum;
    double* pb = P.pre_bias + 6 * (size_t)f;
    pb[0] = ba.x; pb[1] = ba.y; pb[2] = ba.z; pb[3] = bg.x; pb[4] = bg.y; pb[5] = bg.z;
  }
  for (int k = lane; k < 225; k += 64) { P.pre_J[225 * (size_t)f + k] = J[k]; P.pre_P[225 * (size_t)f + k] = C[k]; }
  // whitening W = chol(C)^-1 (lower): ||W r||^2 = r^T C^-1 r without ever forming C^-1 (cond(C) ~ 2e8, A.4).
  // Serial on lane 0 inside LDS (15^3/3 flops); F is reused for the inverse.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double* Wm = P.pre_W + 225 * (size_t)f;
  if (lane == 0) {
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
    for (int k = 0; k < 225; ++k) F[k] = 0.0;
    if (!ok) atomicAdd(P.flag + 1, 1);  // factor carries no weight (covariance not positive definite, e.g. no samples): counted, reported in covgpu_result
    if (ok)
      for (int c = 0; c < 15; ++c) {
        F[16 * c] = 1.0 / C[16 * c];
        for (int r = c + 1; r < 15; ++r) {
          double s2 = 0.0;
          for (int k = c; k < r; ++k) s2 += C[15 * r + k] * F[15 * k + c];
          F[15 * r + c] = -s2 / C[16 * r];
        }
      }
    for (int k = 0; k < 225; ++k) Wm[k] = F[k];
  }
}

__global__ __launch_bounds__(64 * kImuWaves) void k_imu_build(DevProblem P) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * kImuWaves + wave;
  const bool live = f < P.I;
  double* sl = sm[wave];
  imu_stage<true>(P, P.pose, P.sb, f, live, lane, sl);
  {
    double c2 = (live && lane < 15) ? 0.5 * sl[1141 + lane] * sl[1141 + lane] : 0.0;
    c2 = wave_sum(c2);
    part_put(P, SC_COST, P.part_imu + blockIdx.x * kImuWaves + wave, c2);
  }
  if (!live) return;
  const double* Jw = sl + 450;
  const double* r = sl + 1141;
  const int i = P.imu_i[f], j = P.imu_j[f];
  // Deterministic scatter of J^T J (30x30; groups P_i(0:6) S_i(6:15) P_j(15:21) S_j(21:30)) and J^T r:
  // every destination has exactly one writer. Blocks that two factors share (a keyframe is the successor of one
  // factor and the predecessor of the next) go to role-indexed slots [role][position] summed by k_imu_gather in a
  // fixed order; the cross blocks (pos_j, pos_i) belong to this factor alone. By construction pj == pi + 1.
  const int pi = P.perm[i], pj = P.perm[j];
  const size_t K = (size_t)P.K;
  if (lane < 30) {
    const int a = lane, role = a / 15, la = a % 15;
    double ga = 0.0, haa = 0.0;
    for (int k = 0; k < 15; ++k) { ga += Jw[30 * k + a] * r[k]; haa += Jw[30 * k + a] * Jw[30 * k + a]; }
    double* g = P.imuG + (role * K + (role ? pj : pi)) * 30;
    g[la] = ga; g[15 + la] = haa;
  }
  for (int e = lane; e < 900; e += 64) {
    const int a = e / 30, b = e - 30 * a;
    const int ga_ = a / 15, la = a % 15, gb_ = b / 15, lb = b % 15;
    double h = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) h += Jw[30 * k + a] * Jw[30 * k + b];
    const size_t slot = ga_ * K + (ga_ ? pj : pi);
    if (la < 6 && lb < 6) {          // pose-pose
      if (ga_ == gb_) P.imuCd[slot * 36 + 6 * la + lb] = h;
      else if (ga_ == 1) P.imuCd[(2 * K + pj) * 36 + 6 * la + lb] = h;   // cross block (pos_j, pos_i): plane 2, added by k_imu_gather after k_pair_blocks
    } else if (la >= 6 && lb >= 6) { // sb-sb
      if (ga_ == gb_) P.imuAd[slot * 81 + 9 * (la - 6) + (lb - 6)] = h;
      else if (ga_ == 1) P.Ae[(size_t)81 * pj + 9 * (la - 6) + (lb - 6)] = h;
    } else if (la >= 6) {            // sb (row) x pose (col)
      if (ga_ == gb_) P.imuBs[slot * 54 + 6 * (la - 6) + lb] = h;
      else if (ga_ == 0) P.Bn[(size_t)54 * pi + 6 * (la - 6) + lb] = h;   // sb_i x pose_j (next)
      else P.Bp[(size_t)54 * pj + 6 * (la - 6) + lb] = h;                 // sb_j x pose_i (previous)
    }
  }
}

// sums the two role slots of every chain position in a fixed order: Ad, Bs, diagonal block of C, gradient, rhs, diag(J^T J)
// which: 1 = everything that lives on speed-bias dimensions (only IMU factors touch those, so it can run before the
// landmark pass and release the chain factorisation early), 0 = the pose-dimension part (adds onto what k_kf_reduce
// assigned: must run after it), 2 = both.
__global__ __launch_bounds__(256) void k_imu_gather(DevProblem P, int which) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int pos = t / 201, e = t - 201 * pos;
  if (pos >= P.K) return;
  const size_t K = (size_t)P.K, s0 = (size_t)pos, s1 = K + pos;
  if (e < 81) {
    if (which != 0) P.Ad[81 * s0 + e] = P.imuAd[81 * s0 + e] + P.imuAd[81 * s1 + e];
  } else if (e < 135) {
    const int q = e - 81;
    if (which != 0) P.Bs[54 * s0 + q] = P.imuBs[54 * s0 + q] + P.imuBs[54 * s1 + q];
  } else if (e < 171) {
    const int q = e - 135, r = q / 6, c = q - 6 * r;
    if (which != 1) {
      if (c <= r) *c_entry(P, pos, pos, r, c) += P.imuCd[36 * s0 + q] + P.imuCd[36 * s1 + q];
      const double x = P.imuCd[36 * (2 * K + pos) + q];  // pose_pos x pose_(pos-1) of the factor ending here (exactly 0: none)
      if (x != 0.0) *c_entry(P, pos, pos - 1, r, c) += x;
    }
  } else {
    const int q = e - 171, kf = P.pos_kf[pos];
    const int dim = q < 15 ? q : q - 15;
    if (which != 2 && (which == 0) != (dim < 6)) return;
    const double v = P.imuG[30 * s0 + q] + P.imuG[30 * s1 + q];
    if (q < 15) { P.grad[(size_t)15 * kf + q] += v; P.bred[(size_t)15 * kf + q] -= v; }
    else P.hdiag[(size_t)15 * kf + q - 15] += v;
  }
}

__global__ __launch_bounds__(64 * kImuWaves) void k_imu_jvp(DevProblem P, const double* __restrict__ v_all) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * kImuWaves + wave;
  const bool live = f < P.I;
  double* sl = sm[wave];
  imu_stage<true>(P, P.pose, P.sb, f, live, lane, sl);
  double acc = 0.0;
  if (live && lane < 15) {
    const double* Jw = sl + 450;
    const double* vi = v_all + 15 * (size_t)P.imu_i[f];
    const double* vj = v_all + 15 * (size_t)P.imu_j[f];
    double s2 = 0.0;
    for (int c = 0; c < 15; ++c) s2 += Jw[30 * lane + c] * vi[c] + Jw[30 * lane + 15 + c] * vj[c];
    acc = s2 * s2;
  }
  acc = wave_sum(acc);
  part_put(P, SC_JV2, P.part_imu + blockIdx.x * kImuWaves + wave, acc);
}

__global__ __launch_bounds__(64 * kImuWaves) void k_imu_cost(DevProblem P, const double* __restrict__ pose, const double* __restrict__ sb) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * kImuWaves + wave;
  const bool live = f < P.I;
  double* sl = sm[wave];
  imu_stage<false>(P, pose, sb, f, live, lane, sl);
  double acc = 0.0;
  if (live && lane < 15) { const double rv = sl[1141 + lane]; acc = 0.5 * rv * rv; }
  acc = wave_sum(acc);
  part_put(P, SC_COST, P.part_imu + blockIdx.x * kImuWaves + wave, acc);
}

__global__ __launch_bounds__(64 * kImuWaves) void k_imu_linearize(DevProblem P, double* r_out, double* J_out) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * kImuWaves + wave;
  const bool live = f < P.I;
  double* sl = sm[wave];
  imu_stage<true>(P, P.pose, P.sb, f, live, lane, sl);
  if (!live) return;
  if (lane < 15) r_out[15 * (size_t)f + lane] = sl[1141 + lane];
  for (int k = lane; k < 450; k += 64) J_out[450 * (size_t)f + k] = sl[450 + k];
}

static inline dim3 imu_grid(int I) { return dim3((I + kImuWaves - 1) / kImuWaves); }

void launch_preintegrate(const DevProblem& P, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_preintegrate, imu_grid(P.I), dim3(64 * kImuWaves), 0, st, P);
}
void launch_imu_build(const DevProblem& P, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_build, imu_grid(P.I), dim3(64 * kImuWaves), 0, st, P);
}
void launch_imu_gather(const DevProblem& P, int which, hipStream_t st) {
  if (!P.vi) return;
  hipLaunchKernelGGL(k_imu_gather, dim3((201 * P.K + 255) / 256), dim3(256), 0, st, P, which);
}
void launch_imu_jvp(const DevProblem& P, const double* v_all, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_jvp, imu_grid(P.I), dim3(64 * kImuWaves), 0, st, P, v_all);
}
void launch_imu_cost(const DevProblem& P, const double* pose, const double* sb, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_cost, imu_grid(P.I), dim3(64 * kImuWaves), 0, st, P, pose, sb);
}
void launch_imu_linearize(const DevProblem& P, double* r, double* J, hipStream_t st) {
  if (P.I == 0) return;
  hipLaunchKernelGGL(k_imu_linearize, imu_grid(P.I), dim3(64 * kImuWaves), 0, st, P, r, J);
}

}  // namespace covgpu
