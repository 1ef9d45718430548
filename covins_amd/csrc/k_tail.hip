// k_tail.hip — the trust-region tail of one iteration in eight launches instead of twenty-five.
//
// Replaces, like k_dense.hip, the step logic and vector arithmetic of ceres::Solve's TrustRegionMinimizer + DoglegStrategy /
// LevenbergMarquardtStrategy (optimization_be.cpp:560-567). After the linear solve an iteration used to be a
// chain of ~25 dependent launches of 5-27 us each (two J*v passes of three kernels each, three cost kernels, seven partial-sum
// finishers, four one-thread logic kernels: ~0.35 ms of a 3.9 ms iteration on the 5-agent map). What makes it short here:
//   * the dogleg step is a LINEAR COMBINATION of two vectors known right after the solve — the scaled gradient c = g / d^2
//     (Cauchy direction) and the Gauss-Newton step n:  step = cg c + cn n.  One pass over the residuals forms
//     JA = |J c|^2, JB = |J n|^2, JC = (J c).(J n); then |J step|^2 = cg^2 JA + 2 cg cn JC + cn^2 JB for ANY (cg, cn): the second
//     J*v pass (after the coefficients were known) is gone, and a rejected step needs none at all. The same for
//     g.step = cg GG + cn GDOT and |step|^2 = cg^2 VV + 2 cg cn VG + cn^2 NN from dot products of the same pass over the vectors;
//   * the three residual families (reprojection | IMU | between) are SEGMENTS of one launch (blockIdx ranges), for J*v and for the cost;
//   * every partial-sum finisher is one launch over all its slots, and the one-thread step logic rides in its last workgroup
//     (ticket counter: nobody waits for anybody). A sharded solve puts its scalar all-reduce between the two instead.
// Deterministic like the kernels it replaces: per-wave partial sums in fixed slots, fixed-order finish.
#include "between_dev.hpp"
#include "common.hpp"
#include "dev_math.hpp"
#include "inertial_dev.hpp"
#include "reduce.hpp"
#include "visual_dev.hpp"

namespace covgpu {
using namespace covdev;

static inline int tail_grid(int n, int cap) {
  const int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > cap ? cap : b);
}
COV_DEV void tail_put(const DevProblem& P, double v, int slot, int index) {
  v = wave_sum(v);
  part_put(P, slot, index, v);
}

// ---- T1: c = g / d^2 (-> vtmp) and every dot product of the step logic: GG = |g/d|^2, GN2 = |d n|^2, GDOT = g.n, GMAX = max|g|,
//          VV = |c|^2, VG = c.n, NN = |n|^2, XN2 = |x|^2 (the state). At most 256 workgroups (one atomic max per wave).
__global__ __launch_bounds__(256) void k_tail_stats(DevProblem P) {
  double gg = 0, gn2 = 0, gd = 0, gm = 0, vv = 0, vg = 0, nn = 0, xn = 0;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int q = t0; q < P.N; q += stride) {
    const double g = P.grad[q], d = clamp_diag(P.hdiag[q]), s = P.gn[q];
    const double w = P.vw ? P.vw[q] : 1.0;  // sharded solve: every unknown is counted by exactly one rank
    const double c = g / (d * d);
    P.vtmp[q] = c;
    gg += w * (g / d) * (g / d); gn2 += w * (d * s) * (d * s); gd += w * g * s; gm = fmax(gm, w * fabs(g));
    vv += w * c * c; vg += w * c * s; nn += w * s * s;
  }
  for (int t = t0; t < P.K; t += stride) {
    const double wp = P.vw ? P.vw[(size_t)P.D * t] : 1.0, ws = (P.vw && P.vi) ? P.vw[(size_t)P.D * t + 6] : 1.0;
    if (!P.fixed[t]) for (int k = 0; k < 7; ++k) xn += wp * P.pose[7 * t + k] * P.pose[7 * t + k];
    if (P.vi) for (int k = 0; k < 9; ++k) xn += ws * P.sb[9 * t + k] * P.sb[9 * t + k];
  }
  for (int q = t0; q < 3 * P.L; q += stride) xn += P.lm[q] * P.lm[q];
  const int idx = P.part_vec + blockIdx.x * 4 + (threadIdx.x >> 6);
  tail_put(P, gg, SC_GG, idx); tail_put(P, gn2, SC_GN2, idx); tail_put(P, gd, SC_GDOT, idx);
  tail_put(P, vv, SC_VV, idx); tail_put(P, vg, SC_VG, idx); tail_put(P, nn, SC_NN, idx); tail_put(P, xn, SC_XN2, idx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(&P.scal[SC_GMAX]), (unsigned long long)__double_as_longlong(gm));  // (order-independent; cleared with the system)
}

// ---- T2: JA = |J va|^2, JB = |J vb|^2, JC = (J va).(J vb) over all residual blocks, at the current estimate. TWO == false: vb only (LM).
//          Segments by blockIdx: [0, nb_obs) reprojection (grid-stride over the SoA stream) | [nb_obs, nb_obs + nb_imu) one wave per
//          IMU factor | the rest: one thread per between factor.
static_assert(kImuWaves == 4, "k_tail_jvp / k_tail_cost index their IMU and edge segments as 4 waves / 256 threads per workgroup");
template <bool TWO>
__global__ __launch_bounds__(64 * kImuWaves) void k_tail_jvp(DevProblem P, const double* __restrict__ va, const double* __restrict__ vb, int nb_obs, int nb_imu) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double ja = 0.0, jb = 0.0, jc = 0.0;
  int idx;
  if ((int)blockIdx.x < nb_obs) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < P.O; o += nb_obs * blockDim.x) {
      ObsLin e;
      const int kf = P.obs_kf[o], l = P.obs_lm[o];
      eval_obs<true>(P, P.pose, P.lm, o, kf, l, e);
      {
        const double* vp = vb + (size_t)P.D * kf;
        const double* vl = vb + P.n + 3 * (size_t)l;
        double s0 = e.jl[0] * vl[0] + e.jl[1] * vl[1] + e.jl[2] * vl[2];
        double s1 = e.jl[3] * vl[0] + e.jl[4] * vl[1] + e.jl[5] * vl[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) { s0 += e.jp[k] * vp[k]; s1 += e.jp[6 + k] * vp[k]; }
        jb += s0 * s0 + s1 * s1;
        if (TWO) {
          const double* up = va + (size_t)P.D * kf;
          const double* ul = va + P.n + 3 * (size_t)l;
          double t0 = e.jl[0] * ul[0] + e.jl[1] * ul[1] + e.jl[2] * ul[2];
          double t1 = e.jl[3] * ul[0] + e.jl[4] * ul[1] + e.jl[5] * ul[2];
#pragma unroll
          for (int k = 0; k < 6; ++k) { t0 += e.jp[k] * up[k]; t1 += e.jp[6 + k] * up[k]; }
          ja += t0 * t0 + t1 * t1; jc += t0 * s0 + t1 * s1;
        }
      }
    }
    idx = blockIdx.x * 4 + wave;
  } else if ((int)blockIdx.x < nb_obs + nb_imu) {
    const int bi = blockIdx.x - nb_obs;
    const int f = bi * kImuWaves + wave;
    const bool live = f < P.I;
    double* sl = sm[wave];
    imu_stage<true>(P, P.pose, P.sb, f, live, lane, sl);
    if (live && lane < 15) {
      const double* Jw = sl + 450;
      const size_t oi = 15 * (size_t)P.imu_i[f], oj = 15 * (size_t)P.imu_j[f];
      double s2 = 0.0, t2 = 0.0;
      for (int c = 0; c < 15; ++c) {
        s2 += Jw[30 * lane + c] * vb[oi + c] + Jw[30 * lane + 15 + c] * vb[oj + c];
        if (TWO) t2 += Jw[30 * lane + c] * va[oi + c] + Jw[30 * lane + 15 + c] * va[oj + c];
      }
      jb = s2 * s2; ja = t2 * t2; jc = s2 * t2;
    }
    idx = P.part_imu + bi * kImuWaves + wave;
  } else {
    const int be = blockIdx.x - nb_obs - nb_imu;
    const int e = be * 256 + threadIdx.x;
    if (e < P.E) {
      double r[6], J[72];
      eval_edge<true>(P, P.pose, e, r, J);
      const size_t oi = (size_t)P.D * P.edge_i[e], oj = (size_t)P.D * P.edge_j[e];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          s += J[12 * a + c] * vb[oi + c] + J[12 * a + 6 + c] * vb[oj + c];
          if (TWO) t += J[12 * a + c] * va[oi + c] + J[12 * a + 6 + c] * va[oj + c];
        }
        jb += s * s; ja += t * t; jc += s * t;
      }
    }
    idx = P.part_edge + be * 4 + wave;
    if (be * 4 + wave >= (P.E + 63) / 64) idx = -1;   // (the edge range of a slot holds one entry per 64 edges)
  }
  jb = wave_sum(jb);
  if (TWO) { ja = wave_sum(ja); jc = wave_sum(jc); }
  if (idx >= 0) {
    part_put(P, SC_JB, idx, jb);
    if (TWO) { part_put(P, SC_JA, idx, ja); part_put(P, SC_JC, idx, jc); }
  }
}

// ---- T5: cost of the candidate estimate, the same three segments
__global__ __launch_bounds__(64 * kImuWaves) void k_tail_cost(DevProblem P, int nb_obs, int nb_imu) {
  __shared__ double sm[kImuWaves][kImuLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  int idx;
  if ((int)blockIdx.x < nb_obs) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < P.O; o += nb_obs * blockDim.x) {
      ObsLin e;
      eval_obs<false>(P, P.pose_c, P.lm_c, o, P.obs_kf[o], P.obs_lm[o], e);
      acc += e.cost;
    }
    idx = blockIdx.x * 4 + wave;
  } else if ((int)blockIdx.x < nb_obs + nb_imu) {
    const int bi = blockIdx.x - nb_obs;
    const int f = bi * kImuWaves + wave;
    const bool live = f < P.I;
    double* sl = sm[wave];
    imu_stage<false>(P, P.pose_c, P.sb_c, f, live, lane, sl);
    if (live && lane < 15) { const double rv = sl[1141 + lane]; acc = 0.5 * rv * rv; }
    idx = P.part_imu + bi * kImuWaves + wave;
  } else {
    const int be = blockIdx.x - nb_obs - nb_imu;
    const int e = be * 256 + threadIdx.x;
    if (e < P.E) { double r[6]; acc = eval_edge<false>(P, P.pose_c, e, r, nullptr); }
    idx = P.part_edge + be * 4 + wave;
    if (be * 4 + wave >= (P.E + 63) / 64) idx = -1;
  }
  acc = wave_sum(acc);
  if (idx >= 0) part_put(P, SC_COST, idx, acc);
}

// ---- step logic (one thread). Stage A = what k_tr_after_solve + k_combine_step's scalars + k_tr_after_model decided, from the
//      dot products of T1 / T2; stage B = k_tr_decide. `ld` reads a reduced scalar.
struct ScalRead {
  const double* p; bool dev_scope;   // dev_scope: written by other workgroups of THIS launch (device-scope accesses, no fence)
  COV_DEV double operator()(int i) const { return dev_scope ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i]; }
};
COV_DEV void tail_logic_a(const DevProblem& P, const TrConsts& tc, int fresh, const ScalRead& sc, int flag0) {
  double* t = P.tr;
  t[TR_RETRY] = 0.0; t[TR_VALID] = 0.0; t[TR_ACC] = 0.0; t[TR_FNCONV] = 0.0; t[TR_MODEL] = 0.0; t[TR_SN] = 0.0;
  if (fresh) {
    if (t[TR_FIRST] != 0.0) { t[TR_COST] = sc(SC_COST); t[TR_INITCOST] = sc(SC_COST); t[TR_FIRST] = 0.0; }
    const bool ok = flag0 == 0;
    t[TR_OK] = ok ? 1.0 : 0.0;
    if (!ok && tc.strategy == COVGPU_DOGLEG && t[TR_MU] * 10.0 < 1.0) { t[TR_MU] *= 10.0; t[TR_RETRY] = 1.0; return; }  // ComputeGaussNewtonStep: raise mu, solve again
    if (!ok && tc.strategy == COVGPU_DOGLEG) t[TR_MU] *= 10.0;
    if (sc(SC_GMAX) <= tc.gradient_tolerance) { t[TR_TERM] = 3.0; return; }
    t[TR_GG] = sc(SC_GG); t[TR_GN2] = sc(SC_GN2); t[TR_GDOT] = sc(SC_GDOT);
    t[TR_JA] = sc(SC_JA); t[TR_JB] = sc(SC_JB); t[TR_JC] = sc(SC_JC); t[TR_VV] = sc(SC_VV); t[TR_VG] = sc(SC_VG); t[TR_NN] = sc(SC_NN);
    t[TR_XN2] = sc(SC_XN2);
    if (tc.strategy == COVGPU_DOGLEG) t[TR_ALPHA] = t[TR_GG] / t[TR_JA];
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
  // model decrease and step norm of step = cg c + cn n (k_tr_after_model)
  if (t[TR_TERM] != 0.0) return;
  const bool ok = t[TR_OK] != 0.0;
  const double gs = cg * t[TR_GG] + cn * t[TR_GDOT];
  const double jv2 = cg * cg * t[TR_JA] + 2.0 * cg * cn * t[TR_JC] + cn * cn * t[TR_JB];
  const double sn2 = cg * cg * t[TR_VV] + 2.0 * cg * cn * t[TR_VG] + cn * cn * t[TR_NN];
  const double model = ok ? -(gs + 0.5 * jv2) : 0.0, sn = ok ? sqrt(fmax(sn2, 0.0)) : 0.0;
  t[TR_MODEL] = model; t[TR_SN] = sn;
  const bool valid = ok && model > 0.0;
  t[TR_VALID] = valid ? 1.0 : 0.0;
  if (valid && sn <= tc.parameter_tolerance * (sqrt(t[TR_XN2]) + tc.parameter_tolerance)) t[TR_TERM] = 2.0;
}
COV_DEV void tail_logic_b(const DevProblem& P, const TrConsts& tc, const ScalRead& sc) {
  double* t = P.tr;
  if (t[TR_RETRY] != 0.0 || t[TR_TERM] != 0.0) return;
  const bool lm = tc.strategy == COVGPU_LM;
  int acc = 0;
  if (t[TR_VALID] == 0.0) {  // invalid step: failed factorisation or no model decrease
    if (lm) { t[TR_RADIUS] /= t[TR_LMDF]; t[TR_LMDF] *= 2.0; }
    else t[TR_MU] *= 10.0;
    t[TR_REUSE] = 0.0;
    if (t[TR_MU] >= 1.0 && t[TR_OK] == 0.0) t[TR_TERM] = 4.0;
  } else {
    const double cost = t[TR_COST], cost_new = sc(SC_COST);
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

// ---- T3 / T6: fixed-order sums of the listed slots' partials -> scal[slot], one workgroup per slot; logic != 0: the workgroup that
//      finishes LAST (ticket in flag[2]) runs stage A (logic 1) or B (logic 2) of the step logic on the sums of all of them.
struct SlotList { int n; int slot[12]; };
__global__ __launch_bounds__(1024) void k_tail_finish(DevProblem P, SlotList sl, TrConsts tc, int logic, int fresh) {
  __shared__ double sc[1024];
  __shared__ int s_ticket;
  const int slot = sl.slot[blockIdx.x];
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
  if (logic == 0) { if (threadIdx.x == 0) P.scal[slot] = sc[0]; return; }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&P.scal[slot], sc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sum is out before the ticket is drawn (k_bwd_front uses the same idiom)
    // (release on the ticket, acquire for the winner — inside the HIP memory model, not only what gfx9 happens to do; this kernel
    //  runs alone on the chip, so the cache maintenance the release / acquire imply costs nothing measurable)
    s_ticket = __hip_atomic_fetch_add(&P.flag[2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != (int)gridDim.x - 1 || threadIdx.x != 0) return;
  __hip_atomic_store(&P.flag[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (next launch: ordered by the stream)
  const ScalRead rd{P.scal, true};
  if (logic == 1) tail_logic_a(P, tc, fresh, rd, __hip_atomic_load(&P.flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else tail_logic_b(P, tc, rd);
}
// the step logic alone: a rejected step re-derives its coefficients from the stored dot products (stage A, fresh = 0); a sharded
// solve runs it after the scalar all-reduce (reads the reduced copies)
__global__ void k_tail_logic(DevProblem P, TrConsts tc, int stage, int fresh) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const ScalRead rd{P.scal_r, false};
  if (stage == 1) tail_logic_a(P, tc, fresh, rd, P.flag_r[0]);
  else tail_logic_b(P, tc, rd);
}

// ---- T4: candidate = x (+) (cg c + cn n)   (R1: q+ = q (x) Exp(dtheta), renormalised; p+ = p + dp; plain addition elsewhere)
__global__ __launch_bounds__(256) void k_tail_apply(DevProblem P) {
  const double cg = P.tr[TR_CG], cn = P.tr[TR_CN];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  auto st = [&](size_t q) { const double s = cn * P.gn[q]; return cg != 0.0 ? s + cg * P.vtmp[q] : s; };
  if (t < P.K) {
    const size_t b = (size_t)P.D * t;
    const double* x = P.pose + 7 * t;
    double* y = P.pose_c + 7 * t;
    if (P.fixed[t]) {
      for (int k = 0; k < 7; ++k) y[k] = x[k];
    } else {
      const Q4 q = qnormalize(qmul(ldq(x), qexp(V3{st(b), st(b + 1), st(b + 2)})));
      y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w;
      y[4] = x[4] + st(b + 3); y[5] = x[5] + st(b + 4); y[6] = x[6] + st(b + 5);
    }
    if (P.vi)
      for (int k = 0; k < 9; ++k) P.sb_c[9 * t + k] = P.sb[9 * t + k] + st(b + 6 + k);
  }
  for (int q = t; q < 3 * P.L; q += gridDim.x * blockDim.x) P.lm_c[q] = P.lm[q] + st((size_t)P.n + q);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline void tail_segments(const DevProblem& P, int& nb_obs, int& nb_imu, int& nb_edge) {
  nb_obs = P.O > 0 ? tail_grid(P.O, 2048) : 0;   // (<= 8192 wave slots: the observation range of a partial-sum slot)
  nb_imu = (P.vi && P.I > 0) ? (P.I + kImuWaves - 1) / kImuWaves : 0;
  nb_edge = P.E > 0 ? (P.E + 255) / 256 : 0;
}
void launch_tail_stats(const DevProblem& P, hipStream_t st) {
  hipLaunchKernelGGL(k_tail_stats, dim3(tail_grid(P.N, 256)), dim3(256), 0, st, P);
}
void launch_tail_jvp(const DevProblem& P, bool two, hipStream_t st) {
  int a, b, c;
  tail_segments(P, a, b, c);
  if (a + b + c == 0) return;
  if (two) hipLaunchKernelGGL(k_tail_jvp<true>, dim3(a + b + c), dim3(64 * kImuWaves), 0, st, P, (const double*)P.vtmp, (const double*)P.gn, a, b);
  else hipLaunchKernelGGL(k_tail_jvp<false>, dim3(a + b + c), dim3(64 * kImuWaves), 0, st, P, (const double*)nullptr, (const double*)P.gn, a, b);
}
void launch_tail_cost(const DevProblem& P, hipStream_t st) {
  int a, b, c;
  tail_segments(P, a, b, c);
  if (a + b + c == 0) return;
  hipLaunchKernelGGL(k_tail_cost, dim3(a + b + c), dim3(64 * kImuWaves), 0, st, P, a, b);
}
// stage 1: the sums of T1 / T2 (+ step logic A if `with_logic`); stage 2: the candidate cost (+ step logic B)
void launch_tail_finish(const DevProblem& P, TrConsts tc, int stage, bool two, bool with_logic, int fresh, hipStream_t st) {
  SlotList sl;
  sl.n = 0;
  auto add = [&](int s) { sl.slot[sl.n++] = s; };
  if (stage == 1) {
    add(SC_GG); add(SC_GN2); add(SC_GDOT); add(SC_VV); add(SC_VG); add(SC_NN); add(SC_XN2); add(SC_JB);
    if (two) { add(SC_JA); add(SC_JC); }
  } else add(SC_COST);
  for (int i = sl.n; i < 12; ++i) sl.slot[i] = 0;
  hipLaunchKernelGGL(k_tail_finish, dim3(sl.n), dim3(1024), 0, st, P, sl, tc, with_logic ? stage : 0, fresh);
}
void launch_tail_logic(const DevProblem& P, TrConsts tc, int stage, int fresh, hipStream_t st) {
  hipLaunchKernelGGL(k_tail_logic, dim3(1), dim3(64), 0, st, P, tc, stage, fresh);
}
void launch_tail_apply(const DevProblem& P, hipStream_t st) {
  const int n = P.K > 3 * P.L ? P.K : 3 * P.L;
  int g = (P.K + 255) / 256;
  const int g2 = tail_grid(n, 2048);
  if (g2 > g) g = g2;
  hipLaunchKernelGGL(k_tail_apply, dim3(g), dim3(256), 0, st, P);
}

}  // namespace covgpu
