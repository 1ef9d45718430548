// k_dense.hip — flat vector kernels of the trust-region step (the dense reduced-system solve is k_chol.hip).
//
// Replaces the linear-algebra half of ceres::Solve(SPARSE_SCHUR) (optimization_be.cpp:560-567, 1024-1031):
// Cholesky factorisation of the reduced camera system (CHOLMOD in the reference) and the two triangular
// solves; and the vector arithmetic of Ceres' DoglegStrategy / LevenbergMarquardtStrategy.
//
// Factorisation = right-looking blocked Cholesky on the lower triangle, panel width kTile = 128:
//   potrf_inv  one workgroup factors the 128x128 diagonal block in LDS and also forms L11^-1 there;
//   gemm_abt<TRSM>   A21 <- A21 * L11^-T as an MFMA GEMM against the explicit inverse;
//   gemm_abt<SYRK>   A22 <- A22 - A21 A21^T, one 128x128 tile per workgroup, v_mfma_f64_16x16x4_f64,
//                    4 waves x (4x4) MFMA tiles, K staged through LDS in chunks of 32 with register prefetch.
// Triangular solves reuse the stored L_pp^-1 blocks: one small launch per panel.
#include <algorithm>

#include "common.hpp"
#include "dev_math.hpp"
#include "reduce.hpp"

namespace covgpu {
using namespace covdev;

// ------------------------------------------------------------------------------------------- vector kernels
// add the trust-region damping mu * clamp(diag)^2 to every active diagonal entry of the reduced camera system;
// constant / unconstrained dimensions (diag(J^T J) == 0) become identity rows with zero right-hand side (A.6)
// which: 0 = pose rows (+ the padding rows of the dense pose-graph matrix), 1 = speed-bias rows (in P.Ad, before k_nd_assemble
// copies them into the fronts), 2 = both.
__global__ __launch_bounds__(256) void k_finalize_diag(DevProblem P, double mu, int which) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < P.n) {
    const int kf = q / P.D, r = q - kf * P.D, pos = P.perm[kf];
    if (which == 2 || (which == 0) == (r < 6)) {
      // agent-sharded solve: unknowns of other ranks' subtrees are not this rank's business; the top (replicated) unknowns get
      // their damping after the all-reduce, from the all-reduced diag(J^T J) (k_nd_top_damp, k_front.hip)
      const int own = P.nd_vown != nullptr ? P.nd_vown[2 * pos + (r < 6 ? 0 : 1)] : 1;
      if (own == 1) {
        double* d = (r < 6) ? c_entry(P, pos, pos, r, r) : P.Ad + (size_t)81 * pos + 10 * (r - 6);
        const double h = P.hdiag[q];
        if (h == 0.0) { *d = 1.0; P.bred[q] = 0.0; }
        else { const double c = clamp_diag(h); *d += mu * c * c; }
      }
    }
  }
  const int pad = 6 * P.K + q;  // padding rows of the dense pose-graph matrix (the fronts: k_nd_zero)
  if (which != 1 && !P.nd && q < P.npad - 6 * P.K) P.Sred[(size_t)pad * P.npad + pad] = 1.0;
}

// fixed-order sum of one slot's partials -> scal[slot]
// (1024 threads, four independent partial sums each: with 256 threads walking ~73 partials apiece in one dependent chain
//  this tiny kernel took 20 us, five times per trust-region iteration)
__global__ __launch_bounds__(1024) void k_part_finish(DevProblem P, int slot0) {
  __shared__ double sc[1024];
  const int slot = slot0 + blockIdx.x;
  const double* src = P.part + (size_t)slot * P.part_n;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = threadIdx.x;
  for (; k + 3072 < P.part_n; k += 4096) { a0 += src[k]; a1 += src[k + 1024]; a2 += src[k + 2048]; a3 += src[k + 3072]; }
  for (; k < P.part_n; k += 1024) a0 += src[k];
  sc[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int s2 = 512; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) sc[threadIdx.x] += sc[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) P.scal[slot] = sc[0];
}

COV_DEV void vec_reduce(const DevProblem& P, double v, int slot) {
  v = wave_sum(v);
  part_put(P, slot, P.part_vec + blockIdx.x * 4 + (threadIdx.x >> 6), v);
}
COV_DEV void vec_reduce_max(double v, double* dst) {  // max is order-independent: a plain atomic is deterministic
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v));
}

// GG = |g/d|^2, GN2 = |d gn|^2, GDOT = g.gn, GMAX = max|g|
__global__ __launch_bounds__(256) void k_dogleg_stats(DevProblem P) {
  double gg = 0, gn2 = 0, gd = 0, gm = 0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double g = P.grad[q], d = clamp_diag(P.hdiag[q]), s = P.gn[q];
    const double w = P.vw ? P.vw[q] : 1.0;  // sharded solve: every unknown is counted by exactly one rank
    gg += w * (g / d) * (g / d); gn2 += w * (d * s) * (d * s); gd += w * g * s; gm = fmax(gm, w * fabs(g));
  }
  vec_reduce(P, gg, SC_GG);
  vec_reduce(P, gn2, SC_GN2);
  vec_reduce(P, gd, SC_GDOT);
  vec_reduce_max(gm, &P.scal[SC_GMAX]);
}

__global__ __launch_bounds__(256) void k_cauchy_vec(DevProblem P) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double d = clamp_diag(P.hdiag[q]);
    P.vtmp[q] = P.grad[q] / (d * d);
  }
}

// step = cg * g / d^2 + cn * gn ;  GS = g.step, SN2 = |step|^2
__global__ __launch_bounds__(256) void k_combine_step(DevProblem P, double cg, double cn, int from_dev) {
  if (from_dev) { cg = P.tr[TR_CG]; cn = P.tr[TR_CN]; }
  double gs = 0, sn = 0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double g = P.grad[q], d = clamp_diag(P.hdiag[q]);
    double s = cn * P.gn[q];
    if (cg != 0.0) s += cg * g / (d * d);
    P.step[q] = s;
    const double w = P.vw ? P.vw[q] : 1.0;
    gs += w * g * s; sn += w * s * s;
  }
  vec_reduce(P, gs, SC_GS);
  vec_reduce(P, sn, SC_SN2);
}

// ---- device-side trust region (Ceres 1.x TrustRegionMinimizer + DoglegStrategy / LevenbergMarquardtStrategy) ----
// Same decisions, same order as the host loop they replace (solver.hip keeps that loop for the agent-sharded solve, whose
// scalars pass through the caller's collective): one thread, a few dozen flops, between the vector kernels of the step.
__global__ void k_tr_after_solve(DevProblem P, TrConsts tc, int fresh) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double* t = P.tr;
  t[TR_RETRY] = 0.0; t[TR_VALID] = 0.0; t[TR_ACC] = 0.0; t[TR_FNCONV] = 0.0; t[TR_MODEL] = 0.0; t[TR_SN] = 0.0;
  if (fresh) {
    if (t[TR_FIRST] != 0.0) { t[TR_COST] = P.scal_r[SC_COST]; t[TR_INITCOST] = P.scal_r[SC_COST]; t[TR_FIRST] = 0.0; }
    const bool ok = P.flag_r[0] == 0;
    t[TR_OK] = ok ? 1.0 : 0.0;
    if (!ok && tc.strategy == COVGPU_DOGLEG && t[TR_MU] * 10.0 < 1.0) { t[TR_MU] *= 10.0; t[TR_RETRY] = 1.0; return; }  // ComputeGaussNewtonStep: raise mu, solve again
    if (!ok && tc.strategy == COVGPU_DOGLEG) t[TR_MU] *= 10.0;
    if (P.scal_r[SC_GMAX] <= tc.gradient_tolerance) { t[TR_TERM] = 3.0; return; }
    t[TR_GG] = P.scal_r[SC_GG]; t[TR_GN2] = P.scal_r[SC_GN2]; t[TR_GDOT] = P.scal_r[SC_GDOT];
    if (tc.strategy == COVGPU_DOGLEG) t[TR_ALPHA] = P.scal_r[SC_GG] / P.scal_r[SC_JV2];
  }
  double cg = 0.0, cn = 1.0;
  if (tc.strategy == COVGPU_DOGLEG) {
    const double radius = t[TR_RADIUS], alpha = t[TR_ALPHA], GG = t[TR_GG], GN2 = t[TR_GN2], GDOT = t[TR_GDOT];
    const double gn_norm = sqrt(GN2), g_norm = sqrt(GG);
    if (gn_norm <= radius) { cg = 0.0; cn = 1.0; t[TR_DNORM] = gn_norm; }
    else if (g_norm * alpha >= radius) { cg = -radius / g_norm; cn = 0.0; t[TR_DNORM] = radius; }
    else {
      const double b_dot_a = -alpha * GDOT, a_sq = alpha * alpha * GG;
      const double bma = GN2 - 2.0 * b_dot_a + a_sq, cc = b_dot_a - a_sq;
      const double dd = sqrt(cc * cc + bma * (radius * radius - a_sq));
      const double beta = (cc <= 0.0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
      cg = -alpha * (1.0 - beta); cn = beta; t[TR_DNORM] = radius;
    }
  }
  t[TR_CG] = cg; t[TR_CN] = cn;
}

__global__ void k_tr_after_model(DevProblem P, TrConsts tc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double* t = P.tr;
  if (t[TR_RETRY] != 0.0 || t[TR_TERM] != 0.0) return;
  const bool ok = t[TR_OK] != 0.0;
  const double model = ok ? -(P.scal_r[SC_GS] + 0.5 * P.scal_r[SC_JV2]) : 0.0, sn = ok ? sqrt(P.scal_r[SC_SN2]) : 0.0;
  t[TR_MODEL] = model; t[TR_SN] = sn;
  const bool valid = ok && model > 0.0;
  t[TR_VALID] = valid ? 1.0 : 0.0;
  if (valid && sn <= tc.parameter_tolerance * (sqrt(P.scal_r[SC_XN2]) + tc.parameter_tolerance)) t[TR_TERM] = 2.0;
}

// rho test + updates; k_tr_accept (launched right behind) copies the candidate into the state when the step was accepted
__global__ void k_tr_decide(DevProblem P, TrConsts tc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  {
    double* t = P.tr;
    int acc = 0;
    {
      if (t[TR_RETRY] == 0.0 && t[TR_TERM] == 0.0) {
        const bool lm = tc.strategy == COVGPU_LM;
        if (t[TR_VALID] == 0.0) {  // invalid step: failed factorisation or no model decrease
          if (lm) { t[TR_RADIUS] /= t[TR_LMDF]; t[TR_LMDF] *= 2.0; }
          else t[TR_MU] *= 10.0;
          t[TR_REUSE] = 0.0;
          if (t[TR_MU] >= 1.0 && t[TR_OK] == 0.0) t[TR_TERM] = 4.0;
        } else {
          const double cost = t[TR_COST], cost_new = P.scal_r[SC_COST];
          const double rho = (cost - cost_new) / t[TR_MODEL];
          t[TR_RHO] = rho; t[TR_COSTNEW] = cost_new;
          acc = rho > tc.min_relative_decrease;
          if (acc) {
            t[TR_FNCONV] = fabs(cost - cost_new) <= tc.function_tolerance * cost ? 1.0 : 0.0;
            t[TR_COST] = cost_new;
            if (lm) {
              const double u = 2.0 * rho - 1.0;
              t[TR_RADIUS] = fmin(tc.max_radius, t[TR_RADIUS] / fmax(1.0 / 3.0, 1.0 - u * u * u));
              t[TR_LMDF] = 2.0;
            } else {
              if (rho < 0.25) t[TR_RADIUS] *= 0.5;
              if (rho > 0.75) t[TR_RADIUS] = fmax(t[TR_RADIUS], 3.0 * t[TR_DNORM]);
              t[TR_MU] = fmax(1e-8, 2.0 * t[TR_MU] / 10.0);
            }
            t[TR_REUSE] = 0.0;
          } else if (lm) { t[TR_RADIUS] /= t[TR_LMDF]; t[TR_LMDF] *= 2.0; t[TR_REUSE] = 0.0; }
          else { t[TR_RADIUS] *= 0.5; t[TR_REUSE] = 1.0; }
          if (t[TR_FNCONV] != 0.0) t[TR_TERM] = 1.0;
        }
        t[TR_ACC] = acc ? 1.0 : 0.0;
      }
    }
  }
}
// box != nullptr (round 6): the first workgroup also posts the trust-region state — what the host needs to enqueue the next iteration — into a
// pinned host buffer, the sequence number behind it: the host polls that word instead of sleeping in hipStreamSynchronize behind a D2H copy
__global__ __launch_bounds__(256) void k_tr_accept(DevProblem P, double* box, double seq) {
  if (box != nullptr && blockIdx.x == 0) {
    if (threadIdx.x < TR_COUNT) __hip_atomic_store(box + threadIdx.x, P.tr[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(box + TR_COUNT, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (P.tr[TR_ACC] == 0.0) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t q = t0; q < (size_t)7 * P.K; q += stride) P.pose[q] = P.pose_c[q];
  if (P.vi) for (size_t q = t0; q < (size_t)9 * P.K; q += stride) P.sb[q] = P.sb_c[q];
  for (size_t q = t0; q < (size_t)3 * P.L; q += stride) P.lm[q] = P.lm_c[q];
}

// candidate = x (+) step   (R1: q+ = q (x) Exp(dtheta), renormalised; p+ = p + dp; plain addition elsewhere)
__global__ __launch_bounds__(256) void k_apply_step(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < P.K) {
    const double* s = P.step + (size_t)P.D * t;
    const double* x = P.pose + 7 * t;
    double* y = P.pose_c + 7 * t;
    if (P.fixed[t]) {
      for (int k = 0; k < 7; ++k) y[k] = x[k];
    } else {
      const Q4 q = qnormalize(qmul(ldq(x), qexp(V3{s[0], s[1], s[2]})));
      y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w;
      y[4] = x[4] + s[3]; y[5] = x[5] + s[4]; y[6] = x[6] + s[5];
    }
    if (P.vi)
      for (int k = 0; k < 9; ++k) P.sb_c[9 * t + k] = P.sb[9 * t + k] + s[6 + k];
  }
  for (int q = t; q < 3 * P.L; q += gridDim.x * blockDim.x) P.lm_c[q] = P.lm[q] + P.step[P.n + q];
}

__global__ __launch_bounds__(256) void k_xnorm(DevProblem P) {
  double acc = 0.0;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int t = t0; t < P.K; t += stride) {
    const double wp = P.vw ? P.vw[(size_t)P.D * t] : 1.0, ws = (P.vw && P.vi) ? P.vw[(size_t)P.D * t + 6] : 1.0;
    if (!P.fixed[t]) for (int k = 0; k < 7; ++k) acc += wp * P.pose[7 * t + k] * P.pose[7 * t + k];
    if (P.vi) for (int k = 0; k < 9; ++k) acc += ws * P.sb[9 * t + k] * P.sb[9 * t + k];
  }
  for (int q = t0; q < 3 * P.L; q += stride) acc += P.lm[q] * P.lm[q];
  vec_reduce(P, acc, SC_XN2);
}

static inline int vec_grid(int n) {
  const int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

void launch_finalize_diag(const DevProblem& P, double mu, int which, hipStream_t st) {
  const int cnt = P.n > P.npad ? P.n : P.npad;
  hipLaunchKernelGGL(k_finalize_diag, dim3((cnt + 255) / 256), dim3(256), 0, st, P, mu, which);
}
// One launch clears every small per-iteration buffer: sixteen hipMemsetAsync calls were ~80 us of dispatch latency at the
// head of every linearisation (each fill kernel runs ~5 us whatever its size).
struct ZeroList { double* p[16]; size_t n[16]; int* flag; };
__global__ __launch_bounds__(256) void k_zero_many(ZeroList z) {
  double* p = z.p[blockIdx.y];
  const size_t n = z.n[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.0;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && z.flag != nullptr) *z.flag = 0;   // (the factorisation's failure flag: was a fill launch of its own)
}
void launch_zero_system(const DevProblem& P, hipStream_t st) {
  ZeroList z; int m = 0;
  auto add = [&](double* p, size_t n) { if (p != nullptr && n > 0) { z.p[m] = p; z.n[m] = n; ++m; } };
  add(P.bred, (size_t)P.n);
  if (P.vi) {
    add(P.Ad, (size_t)81 * P.K); add(P.Ae, (size_t)81 * P.K);
    add(P.Bp, (size_t)54 * P.K); add(P.Bs, (size_t)54 * P.K); add(P.Bn, (size_t)54 * P.K);
    add(P.imuAd, (size_t)2 * 81 * P.K); add(P.imuBs, (size_t)2 * 54 * P.K); add(P.imuCd, (size_t)3 * 36 * P.K); add(P.imuG, (size_t)2 * 30 * P.K);
  }
  // (round 6: the keyframe part only — the landmark part of both is ASSIGNED by the landmark linearisation for every landmark, which therefore
  //  depends on nothing this launch clears and goes out first on its own stream: solver.hip enqueue_build)
  add(P.grad, (size_t)P.n); add(P.hdiag, (size_t)P.n);
  add(P.part + (size_t)SC_COST * P.part_n, (size_t)P.part_n);
  add(P.scal + SC_GMAX, 1);   // (k_dogleg_stats takes an atomic max into it)
  for (int i = m; i < 16; ++i) { z.p[i] = nullptr; z.n[i] = 0; }
  z.flag = P.flag;
  hipLaunchKernelGGL(k_zero_many, dim3(128, m), dim3(256), 0, st, z);
}
void launch_part_clear(const DevProblem& P, int slot0, int nslots, hipStream_t st) {
  hipMemsetAsync(P.part + (size_t)slot0 * P.part_n, 0, (size_t)nslots * P.part_n * sizeof(double), st);
}
void launch_part_finish(const DevProblem& P, int slot0, int nslots, hipStream_t st) {
  hipLaunchKernelGGL(k_part_finish, dim3(nslots), dim3(1024), 0, st, P, slot0);
}
void launch_dogleg_stats(const DevProblem& P, hipStream_t st) {
  // (no clearing of the partial sums: every wave of a reduction kernel STORES its slot, the same slots at every launch — the
  //  fill launches were seven ~5 us links of the launch-bound tail of an iteration; SC_GMAX is cleared with the system)
  // (at most 256 workgroups: every wave ends with an atomic max on one address — 4500 of them took 45 us)
  hipLaunchKernelGGL(k_dogleg_stats, dim3(std::min(vec_grid(P.N), 256)), dim3(256), 0, st, P);
  launch_part_finish(P, SC_GG, 3, st);
}
void launch_cauchy_vec(const DevProblem& P, hipStream_t st) {
  hipLaunchKernelGGL(k_cauchy_vec, dim3(vec_grid(P.N)), dim3(256), 0, st, P);
}
void launch_combine_step(const DevProblem& P, double cg, double cn, hipStream_t st) {
  hipLaunchKernelGGL(k_combine_step, dim3(vec_grid(P.N)), dim3(256), 0, st, P, cg, cn, 0);
  launch_part_finish(P, SC_GS, 2, st);
}
void launch_combine_step_dev(const DevProblem& P, hipStream_t st) {
  hipLaunchKernelGGL(k_combine_step, dim3(vec_grid(P.N)), dim3(256), 0, st, P, 0.0, 0.0, 1);
  launch_part_finish(P, SC_GS, 2, st);
}
void launch_tr_after_solve(const DevProblem& P, TrConsts tc, int fresh, hipStream_t st) { hipLaunchKernelGGL(k_tr_after_solve, dim3(1), dim3(64), 0, st, P, tc, fresh); }
void launch_tr_after_model(const DevProblem& P, TrConsts tc, hipStream_t st) { hipLaunchKernelGGL(k_tr_after_model, dim3(1), dim3(64), 0, st, P, tc); }
void launch_tr_accept(const DevProblem& P, hipStream_t st, double* box, double seq) {   // x = candidate if the step logic accepted it (TR_ACC)
  const size_t n = std::max((size_t)9 * P.K, (size_t)3 * P.L);
  hipLaunchKernelGGL(k_tr_accept, dim3(vec_grid((int)std::min<size_t>(n, 1u << 30))), dim3(256), 0, st, P, box, seq);
}
void launch_tr_decide(const DevProblem& P, TrConsts tc, hipStream_t st) {
  hipLaunchKernelGGL(k_tr_decide, dim3(1), dim3(64), 0, st, P, tc);
  launch_tr_accept(P, st);
}
void launch_apply_step(const DevProblem& P, hipStream_t st) {
  const int n = P.K > 3 * P.L ? P.K : 3 * P.L;
  int g = (P.K + 255) / 256;
  const int g2 = vec_grid(n);
  if (g2 > g) g = g2;
  hipLaunchKernelGGL(k_apply_step, dim3(g), dim3(256), 0, st, P);
}
void launch_accept(const DevProblem& P, hipStream_t st) {
  hipMemcpyAsync(P.pose, P.pose_c, (size_t)7 * P.K * sizeof(double), hipMemcpyDeviceToDevice, st);
  if (P.vi) hipMemcpyAsync(P.sb, P.sb_c, (size_t)9 * P.K * sizeof(double), hipMemcpyDeviceToDevice, st);
  if (P.L) hipMemcpyAsync(P.lm, P.lm_c, (size_t)3 * P.L * sizeof(double), hipMemcpyDeviceToDevice, st);
}
void launch_xnorm(const DevProblem& P, hipStream_t st) {
  const int n = P.K > 3 * P.L ? P.K : 3 * P.L;
  hipLaunchKernelGGL(k_xnorm, dim3(vec_grid(n)), dim3(256), 0, st, P);
  launch_part_finish(P, SC_XN2, 1, st);
}

}  // namespace covgpu
