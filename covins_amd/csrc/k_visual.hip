// k_visual.hip — reprojection residual blocks on gfx950: linearisation fused with landmark elimination.
//
// Replaces, for the problems assembled at covins_backend/src/covins_backend/optimization_be.cpp:432-530,
//   R4 robopt::reprojection::GlobalEuclideanReprError<PinholeCamera, RadTan|Equidistant>  (opt_be.cpp:487-525)
//   R5 aslam::PinholeCamera::project3 + distortion Jacobian
//   R6 ceres::CauchyLoss(1.0) corrector                                                   (opt_be.cpp:302,523)
// and the landmark half of Ceres' SPARSE_SCHUR (opt_be.cpp:561): H_ll, g_l, S -= W H_ll^-1 W^T, b += W H_ll^-1 g_l
// (SURVEY.md A.2, A.5, A.6). None of these have a reference source in-tree; the contract is SURVEY.md Appendix A.
//
// Kernel shapes — three deterministic, atomic-free passes:
//   k_lm_lin       one G-lane sub-wave group per landmark (G = 16: tracks average ~10 observations), lane = observation.
//                  Jacobians live in VGPRs only; H_ll / g_l are reduced with shuffle butterflies inside the group. With the
//                  Cholesky factor of the damped inverse, H_ll^-1 = R R^T, ONE 6x3 record per observation goes to HBM:
//                  Z = (Jp^T Jl) R — then W H_ll^-1 W'^T = Z Z'^T for any two observations of the landmark. (Rounds 1-2 wrote
//                  W, Y = W H_ll^-1 and a 39-double pose-side record: 600 B per observation instead of 144.)
//   k_kf_reduce    one wave per keyframe, lanes over its observations: every observation is RE-LINEARISED (pose and landmark
//                  rows are cache resident, the flops are free next to the 312-B record this replaces), D = Jp^T Jp - Z Z^T,
//                  gradient and right-hand side terms summed in a fixed order -> diagonal 6x6 block, gradient, reduced
//                  right-hand side, diag(J^T J).
//   k_pair_blocks  sixteen lanes per covisible keyframe pair (host-built lists, static per problem), lanes over the common
//                  landmarks: C[i,j] = -sum Z_i Z_j^T with plain stores. Every entry has exactly one writer and a fixed
//                  summation order: two solves of the same problem are bit-identical. (The first version accumulated the
//                  6x6 blocks with FP64 atomics: 11.1 ms and run-to-run rounding differences; these three passes take
//                  0.66 ms on the 5-agent map.)
//   obs_*          one thread per observation over the SoA stream (cost, J*v products, test dumps).
// All writes to the pose system go through c_entry (common.hpp): the fronts of the multifrontal solve (nd_entry), or the dense
// matrix of the per-kernel test entry points.
#include <type_traits>

#include "common.hpp"
#include "dev_math.hpp"
#include "reduce.hpp"
#include "visual_dev.hpp"

namespace covgpu {
using namespace covdev;

COV_DEV void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int G>
COV_DEV double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}

// symmetric 3x3 inverse; h = xx xy xz yy yz zz
COV_DEV void inv3sym(const double* h, double* o) {
  const double c00 = h[3] * h[5] - h[4] * h[4], c01 = h[2] * h[4] - h[1] * h[5], c02 = h[1] * h[4] - h[2] * h[3];
  const double det = h[0] * c00 + h[1] * c01 + h[2] * c02;
  const double id = (fabs(det) > 0.0) ? 1.0 / det : 0.0;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (h[0] * h[5] - h[2] * h[2]) * id; o[4] = (h[1] * h[2] - h[0] * h[4]) * id; o[5] = (h[0] * h[3] - h[1] * h[1]) * id;
}

constexpr int kBuildThreads = 256;

// ------------------------------------------------------------------------------------------------------------
// Landmark elimination in three deterministic, atomic-free passes:
//   k_lm_lin      (landmark-major) per landmark H_ll, g_l reduced in-group, damped inverse; per observation a
//                 record {W_a = Jp^T Jl, Y_a = W_a Hinv, D_a = Jp^T Jp - Y_a W_a^T, diag(Jp^T Jp), Jp^T r, Y_a g_l}
//   k_kf_reduce   (keyframe-major) one wave per keyframe sums its observations' records in a fixed order (lane = observation,
//                 partial sums through LDS) -> diagonal block of C, gradient, reduced right-hand side, diag(J^T J)
//   k_pair_blocks (covisible keyframe pair-major) sixteen lanes per pair (i > j in chain-major order), lane = common landmark:
//                 C[i,j] = -sum over common landmarks of Y_i W_j^T, written with plain stores
// Every entry of C is produced by exactly one wave in a fixed summation order: run-to-run bit-identical
// (SURVEY.md §7 "irregular graph": determinism needed for parity tests) and no FP64 atomic contention.
// ------------------------------------------------------------------------------------------------------------
constexpr int kRec = 39;  // per-keyframe sums of k_kf_reduce: D(21, lower row-major) hd(6) gp(6) yg(6)

// lower Cholesky factor of the symmetric positive definite 3x3 h = (xx xy xz yy yz zz): r = (r00 r10 r11 r20 r21 r22); a
// non-positive pivot (never for a damped inverse) gives zeros
COV_DEV void chol3(const double* h, double* r) {
  const double r00 = h[0] > 0.0 ? sqrt(h[0]) : 0.0, i00 = r00 > 0.0 ? 1.0 / r00 : 0.0;
  const double r10 = h[1] * i00, r20 = h[2] * i00;
  const double d1 = h[3] - r10 * r10, r11 = d1 > 0.0 ? sqrt(d1) : 0.0, i11 = r11 > 0.0 ? 1.0 / r11 : 0.0;
  const double r21 = (h[4] - r20 * r10) * i11;
  const double d2 = h[5] - r20 * r20 - r21 * r21, r22 = d2 > 0.0 ? sqrt(d2) : 0.0;
  r[0] = r00; r[1] = r10; r[2] = r11; r[3] = r20; r[4] = r21; r[5] = r22;
}
// Z = W R for W = Jp^T Jl (6x3) and the lower factor R
COV_DEV void z_of(const ObsLin& e, const double* R, double* Z) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double w0 = e.jp[r] * e.jl[0] + e.jp[6 + r] * e.jl[3], w1 = e.jp[r] * e.jl[1] + e.jp[6 + r] * e.jl[4], w2 = e.jp[r] * e.jl[2] + e.jp[6 + r] * e.jl[5];
    Z[3 * r] = w0 * R[0] + w1 * R[1] + w2 * R[3];
    Z[3 * r + 1] = w1 * R[2] + w2 * R[4];
    Z[3 * r + 2] = w2 * R[5];
  }
}

// (compile-time A/B: -DCOVGPU_LMLIN_WAVES=n / -DCOVGPU_PAIR_WAVES=n cap the registers for n waves per SIMD; tools/gpu_ab.sh with COVGPU_LIBRARY)
#ifndef COVGPU_LMLIN_WAVES
#define COVGPU_LMLIN_WAVES 0
#endif
#ifndef COVGPU_PAIR_WAVES
#define COVGPU_PAIR_WAVES 0
#endif
#if COVGPU_LMLIN_WAVES > 0
#define LMLIN_ATTR __attribute__((amdgpu_waves_per_eu(COVGPU_LMLIN_WAVES, COVGPU_LMLIN_WAVES)))
#else
#define LMLIN_ATTR
#endif
#if COVGPU_PAIR_WAVES > 0
#define PAIR_ATTR __attribute__((amdgpu_waves_per_eu(COVGPU_PAIR_WAVES, COVGPU_PAIR_WAVES)))
#else
#define PAIR_ATTR
#endif
template <int G>
__global__ __launch_bounds__(kBuildThreads) LMLIN_ATTR void k_lm_lin(DevProblem P, double mu, DevSignal sig) {
  constexpr int GROUPS = kBuildThreads / G;
  // "everything before this kernel on its stream is complete" (the previous iteration's state update, the preintegration): the side stream's
  // kernels of this pass start from it — published here instead of by a launch in front of the pass's critical path
  if (sig.flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const int l = blockIdx.x * GROUPS + grp;
  const bool lm_ok = l < P.L;
  const int o0 = lm_ok ? P.lm_obs_ptr[l] : 0;
  const int nobs = lm_ok ? P.lm_obs_ptr[l + 1] - o0 : 0;
  const int nchunk = (nobs + G - 1) / G;

  double h[6] = {0, 0, 0, 0, 0, 0}, gl[3] = {0, 0, 0}, cost = 0.0;
  ObsLin e;
  int kf = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int a = c * G + lane;
    if (a < nobs) {
      kf = P.obs_kf[o0 + a];
      eval_obs<true>(P, P.pose, P.lm, o0 + a, kf, l, e);
      h[0] += e.jl[0] * e.jl[0] + e.jl[3] * e.jl[3];
      h[1] += e.jl[0] * e.jl[1] + e.jl[3] * e.jl[4];
      h[2] += e.jl[0] * e.jl[2] + e.jl[3] * e.jl[5];
      h[3] += e.jl[1] * e.jl[1] + e.jl[4] * e.jl[4];
      h[4] += e.jl[1] * e.jl[2] + e.jl[4] * e.jl[5];
      h[5] += e.jl[2] * e.jl[2] + e.jl[5] * e.jl[5];
      gl[0] += e.jl[0] * e.r0 + e.jl[3] * e.r1;
      gl[1] += e.jl[1] * e.r0 + e.jl[4] * e.r1;
      gl[2] += e.jl[2] * e.r0 + e.jl[5] * e.r1;
      cost += e.cost;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) h[k] = group_sum<G>(h[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) gl[k] = group_sum<G>(gl[k]);
  double hi[6];
  {
    double hd[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};
    const double d0 = clamp_diag(h[0]), d1 = clamp_diag(h[3]), d2 = clamp_diag(h[5]);
    hd[0] += mu * d0 * d0; hd[3] += mu * d1 * d1; hd[5] += mu * d2 * d2;
    inv3sym(hd, hi);
  }
  double R[6];
  chol3(hi, R);
  if (lm_ok && lane == 0) {
    double* gg = P.grad + P.n + 3 * (size_t)l;
    double* hh = P.hdiag + P.n + 3 * (size_t)l;
    gg[0] = gl[0]; gg[1] = gl[1]; gg[2] = gl[2];
    hh[0] = h[0]; hh[1] = h[3]; hh[2] = h[5];
    double* hv = P.HllInv + 6 * (size_t)l;
#pragma unroll
    for (int k = 0; k < 6; ++k) hv[k] = hi[k];
    double* rt = P.lmRT + 9 * (size_t)l;  // R (6) | t = R^T g_l (3): what k_kf_reduce needs of this landmark
#pragma unroll
    for (int k = 0; k < 6; ++k) rt[k] = R[k];
    rt[6] = R[0] * gl[0] + R[1] * gl[1] + R[3] * gl[2];
    rt[7] = R[2] * gl[1] + R[4] * gl[2];
    rt[8] = R[5] * gl[2];
  }
  // per-block cost partial (summed in a fixed order by k_cost_finish)
  cost = wave_sum(cost);
  __shared__ double scost[kBuildThreads / 64];
  if ((threadIdx.x & 63) == 0) scost[threadIdx.x >> 6] = cost;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c4 = 0.0;
#pragma unroll
    for (int k = 0; k < kBuildThreads / 64; ++k) c4 += scost[k];
    P.cost_part[blockIdx.x] = c4;
  }

  for (int c = 0; c < nchunk; ++c) {
    const int a = c * G + lane;
    if (a >= nobs) continue;
    if (nchunk > 1) {  // multi-chunk landmark: this chunk's Jacobians were overwritten above
      kf = P.obs_kf[o0 + a];
      eval_obs<true>(P, P.pose, P.lm, o0 + a, kf, l, e);
    }
    double Z[18];
    z_of(e, R, Z);
    double2* zo = reinterpret_cast<double2*>(P.obsZ + 18 * (size_t)P.obs_zpos[o0 + a]);  // 144-byte record, 16-byte aligned, in its keyframe's block
#pragma unroll
    for (int k = 0; k < 9; ++k) zo[k] = double2{Z[2 * k], Z[2 * k + 1]};
  }
}

// cost = sum of the per-block partials in index order (deterministic), added to the cost scalar
__global__ __launch_bounds__(256) void k_cost_finish(DevProblem P, int nparts) {
  __shared__ double sc[256];
  double acc = 0.0;
  for (int k = threadIdx.x; k < nparts; k += 256) acc += P.cost_part[k];
  sc[threadIdx.x] = acc;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) sc[threadIdx.x] += sc[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) P.part[(size_t)SC_COST * P.part_n + 0] = sc[0];
}

// One wave (= one workgroup) per keyframe: fixed-order sum over its observations. Lanes run over the OBSERVATIONS (lane l takes
// observations l, l+64, ...), each re-linearised from the cache-resident pose / landmark rows plus the landmark's factor R and
// t = R^T g_l (72 B); the 39 partial sums of every lane go through LDS and lane k adds column k in lane order.
__global__ __launch_bounds__(64) void k_kf_reduce(DevProblem P) {
  __shared__ double sp[kRec][65];
  const int kf = blockIdx.x, lane = threadIdx.x;
  if (P.fixed[kf]) return;  // constant pose block: rows stay empty, finalize_diag turns them into identity
  const int o0 = P.kf_obs_ptr[kf], o1 = P.kf_obs_ptr[kf + 1];
  double part[kRec];
#pragma unroll
  for (int k = 0; k < kRec; ++k) part[k] = 0.0;
  for (int t = o0 + lane; t < o1; t += 64) {
    const int l = P.kobs_lm[t];
    const double* uvs = P.kobs + 3 * (size_t)t;
    ObsLin e;
    eval_obs_uvs<true>(P, P.pose, P.lm, uvs[0], uvs[1], uvs[2], kf, l, e);
    const double* rt = P.lmRT + 9 * (size_t)l;
    double R[6], Z[18];
#pragma unroll
    for (int k = 0; k < 6; ++k) R[k] = rt[k];
    const double t0 = rt[6], t1 = rt[7], t2 = rt[8];
    z_of(e, R, Z);
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc <= r; ++cc)
        part[q++] += e.jp[r] * e.jp[cc] + e.jp[6 + r] * e.jp[6 + cc] - (Z[3 * r] * Z[3 * cc] + Z[3 * r + 1] * Z[3 * cc + 1] + Z[3 * r + 2] * Z[3 * cc + 2]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      part[21 + r] += e.jp[r] * e.jp[r] + e.jp[6 + r] * e.jp[6 + r];
      part[27 + r] += e.jp[r] * e.r0 + e.jp[6 + r] * e.r1;
      part[33 + r] += Z[3 * r] * t0 + Z[3 * r + 1] * t1 + Z[3 * r + 2] * t2;
    }
  }
#pragma unroll
  for (int k = 0; k < kRec; ++k) sp[k][lane] = part[k];
  __syncthreads();
  double acc = 0.0;
  if (lane < kRec) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int t = 0; t < 64; t += 4) { a0 += sp[lane][t]; a1 += sp[lane][t + 1]; a2 += sp[lane][t + 2]; a3 += sp[lane][t + 3]; }
    acc = (a0 + a1) + (a2 + a3);
  }
  const double gp = __shfl(acc, (lane >= 33 && lane < 39) ? lane - 6 : lane, 64);  // lanes 33..38 hold yg; fetch the matching gp
  const size_t base = (size_t)P.D * kf;
  const int pos = P.perm[kf];
  if (lane < 21) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= lane) ++r;
    const int c = lane - r * (r + 1) / 2;
    *c_entry(P, pos, pos, r, c) = acc;
  } else if (lane < 27) {
    P.hdiag[base + lane - 21] = acc;
  } else if (lane < 33) {
    P.grad[base + lane - 27] = acc;
  } else if (lane < 39) {
    P.bred[base + lane - 33] = acc - gp;
  }
}

// Sixteen lanes per covisible keyframe pair (eight pairs per workgroup): C[i,j] = -sum over the common landmarks of Z_i Z_j^T.
// Lanes run over the common LANDMARKS (lane g takes terms g, g+16, ...: each a 6x3 times 3x6 product from two contiguous
// 144-byte records), the 36 partial sums of every lane go through LDS and lane g adds entries g, g+16, g+32 in lane order.
// (The first version had lane = block entry, 36 of 64 lanes busy, every lane walking all common landmarks in sequence:
// 0.78 ms on the 5-agent map, the longest kernel of the linearisation.)
constexpr int kPairLanes = 16, kPairsPerWg = 8;
constexpr int kPairChunk = 64;   // consecutive workgroups (of kPairsPerWg pairs) that one XCD takes together
__global__ __launch_bounds__(kPairLanes * kPairsPerWg) PAIR_ATTR void k_pair_blocks(DevProblem P, int pair_xcd_order) {
  __shared__ double sp[kPairsPerWg][36][kPairLanes + 1];
  const int grp = threadIdx.x / kPairLanes, g = threadIdx.x % kPairLanes;
  // XCD-aware order (round 5): workgroup b runs on XCD b % 8 (observed dispatch order; placement affects speed only). The pair list is sorted by
  // (row keyframe, column keyframe) in chain order, and the two keyframes' record blocks (60 KB each) are what a pair reads: with consecutive
  // pair groups dealt round-robin every XCD's L2 fetched every keyframe's block — eight copies of the 125 MB record array per pass. Each XCD
  // now takes contiguous CHUNKS of the list: its L2 holds the keyframe blocks around the row keyframes it is working on.
  // (contiguous EIGHTHS of the list were 36 us slower than the round-robin order — 314 against 279 us — although they cut the fetches 4.7x: the
  //  pairs' work is uneven along the list, one XCD finished last. Chunks of kPairChunk consecutive workgroups, dealt round-robin to the XCDs:
  //  a chunk spans ~10 row keyframes, an XCD gets every eighth chunk.)
  const int nblk = (P.npairs + kPairsPerWg - 1) / kPairsPerWg;
  const int kx = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
  const int lb = pair_xcd_order ? ((kx / kPairChunk) * 8 + xcd) * kPairChunk + kx % kPairChunk : (int)blockIdx.x;
  const int p = lb * kPairsPerWg + grp;
  const bool ok = lb < nblk && p < P.npairs;
  const int e0 = ok ? P.pair_ptr[p] : 0, e1 = ok ? P.pair_ptr[p + 1] : 0;
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  // two common landmarks per lane and trip: four 144-byte records in flight instead of two (the pass runs at the latency of these
  // scattered reads); the second one of an odd tail reads the first again with weight zero
  for (int e = e0 + g; e < e1; e += 2 * kPairLanes) {
    const int e2 = e + kPairLanes;
    const bool two = e2 < e1;
    const double2* y = reinterpret_cast<const double2*>(P.obsZ + 18 * (size_t)P.pair_oa[e]);
    const double2* w = reinterpret_cast<const double2*>(P.obsZ + 18 * (size_t)P.pair_ob[e]);
    const double2* y2 = reinterpret_cast<const double2*>(P.obsZ + 18 * (size_t)P.pair_oa[two ? e2 : e]);
    const double2* w2 = reinterpret_cast<const double2*>(P.obsZ + 18 * (size_t)P.pair_ob[two ? e2 : e]);
    double yv[18], wv[18], yu[18], wu[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double2 a2 = y[k], b2 = w[k], c2 = y2[k], d2 = w2[k];
      yv[2 * k] = a2.x; yv[2 * k + 1] = a2.y; wv[2 * k] = b2.x; wv[2 * k + 1] = b2.y;
      yu[2 * k] = c2.x; yu[2 * k + 1] = c2.y; wu[2 * k] = d2.x; wu[2 * k + 1] = d2.y;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[6 * r + c] += yv[3 * r] * wv[3 * c] + yv[3 * r + 1] * wv[3 * c + 1] + yv[3 * r + 2] * wv[3 * c + 2];
    if (two) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[6 * r + c] += yu[3 * r] * wu[3 * c] + yu[3 * r + 1] * wu[3 * c + 1] + yu[3 * r + 2] * wu[3 * c + 2];
    }
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) sp[grp][k][g] = acc[k];
  __syncthreads();
  if (!ok) return;
  const int pi = P.pair_i[p], pj = P.pair_j[p];
  for (int k = g; k < 36; k += kPairLanes) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int t = 0; t < kPairLanes; t += 2) { a0 += sp[grp][k][t]; a1 += sp[grp][k][t + 1]; }
    *c_entry(P, pi, pj, k / 6, k % 6) = -(a0 + a1);
  }
}

// (Measured and dropped, round 4: one workgroup per keyframe i with i's records staged once in LDS (57.6 KB) so that only the partner's
//  records travel per common landmark — bit-identical, half the scattered reads, and SLOWER: 0.52 instead of 0.27 ms. One or two
//  workgroups per CU (77 KB of LDS each) cannot keep enough partner reads in flight; eight waves per CU with every read in flight can.)
// back-substitution: dl = Hinv (-g_l - sum_a W_a^T dp[kf_a]); written to out_all[n + 3l ..]
template <int G>
__global__ __launch_bounds__(kBuildThreads) void k_lm_backsub(DevProblem P, const double* __restrict__ dp, double* __restrict__ out_all) {
  constexpr int GROUPS = kBuildThreads / G;
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const int l = blockIdx.x * GROUPS + grp;
  const bool lm_ok = l < P.L;
  const int o0 = lm_ok ? P.lm_obs_ptr[l] : 0;
  const int nobs = lm_ok ? P.lm_obs_ptr[l + 1] - o0 : 0;
  double t[3] = {0, 0, 0};
  for (int a = lane; a < nobs; a += G) {
    ObsLin e;
    const int kf = P.obs_kf[o0 + a];
    eval_obs<true>(P, P.pose, P.lm, o0 + a, kf, l, e);
    const double* d = dp + (size_t)P.D * kf;
    // (W^T d)_c = sum_r (Jp^T Jl)[r][c] d[r] = sum_k Jl[k][c] (Jp[k] . d)
    const double s0 = e.jp[0] * d[0] + e.jp[1] * d[1] + e.jp[2] * d[2] + e.jp[3] * d[3] + e.jp[4] * d[4] + e.jp[5] * d[5];
    const double s1 = e.jp[6] * d[0] + e.jp[7] * d[1] + e.jp[8] * d[2] + e.jp[9] * d[3] + e.jp[10] * d[4] + e.jp[11] * d[5];
    t[0] += e.jl[0] * s0 + e.jl[3] * s1;
    t[1] += e.jl[1] * s0 + e.jl[4] * s1;
    t[2] += e.jl[2] * s0 + e.jl[5] * s1;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = group_sum<G>(t[k]);
  if (lm_ok && lane == 0) {
    const double* gl = P.grad + P.n + 3 * (size_t)l;
    const double* hi = P.HllInv + 6 * (size_t)l;
    const double x = -gl[0] - t[0], y = -gl[1] - t[1], z = -gl[2] - t[2];
    double* o = out_all + P.n + 3 * (size_t)l;
    o[0] = hi[0] * x + hi[1] * y + hi[2] * z;
    o[1] = hi[1] * x + hi[3] * y + hi[4] * z;
    o[2] = hi[2] * x + hi[4] * y + hi[5] * z;
  }
}

// sum over observations of |Jp v_p[kf] + Jl v_l[lm]|^2
__global__ __launch_bounds__(256) void k_obs_jvp(DevProblem P, const double* __restrict__ v_all) {
  double acc = 0.0;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < P.O; o += gridDim.x * blockDim.x) {
    ObsLin e;
    const int kf = P.obs_kf[o], l = P.obs_lm[o];
    eval_obs<true>(P, P.pose, P.lm, o, kf, l, e);
    const double* vp = v_all + (size_t)P.D * kf;
    const double* vl = v_all + P.n + 3 * (size_t)l;
    double s0 = e.jl[0] * vl[0] + e.jl[1] * vl[1] + e.jl[2] * vl[2];
    double s1 = e.jl[3] * vl[0] + e.jl[4] * vl[1] + e.jl[5] * vl[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) { s0 += e.jp[k] * vp[k]; s1 += e.jp[6 + k] * vp[k]; }
    acc += s0 * s0 + s1 * s1;
  }
  acc = wave_sum(acc);
  part_put(P, SC_JV2, blockIdx.x * 4 + (threadIdx.x >> 6), acc);
}

__global__ __launch_bounds__(256) void k_obs_cost(DevProblem P, const double* __restrict__ pose, const double* __restrict__ lm) {
  double acc = 0.0;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < P.O; o += gridDim.x * blockDim.x) {
    ObsLin e;
    eval_obs<false>(P, pose, lm, o, P.obs_kf[o], P.obs_lm[o], e);
    acc += e.cost;
  }
  acc = wave_sum(acc);
  part_put(P, SC_COST, blockIdx.x * 4 + (threadIdx.x >> 6), acc);
}

__global__ __launch_bounds__(256) void k_obs_linearize(DevProblem P, double* r, double* Jp, double* Jl, double* cost) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= P.O) return;
  ObsLin e;
  eval_obs<true>(P, P.pose, P.lm, o, P.obs_kf[o], P.obs_lm[o], e);
  r[2 * o] = e.r0; r[2 * o + 1] = e.r1; cost[o] = e.cost;
  for (int k = 0; k < 12; ++k) Jp[12 * (size_t)o + k] = e.jp[k];
  for (int k = 0; k < 6; ++k) Jl[6 * (size_t)o + k] = e.jl[k];
}

__global__ __launch_bounds__(256) void k_obs_norms(DevProblem P, double* norms) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= P.O) return;
  ObsLin e;
  eval_obs<false>(P, P.pose, P.lm, o, P.obs_kf[o], P.obs_lm[o], e);
  norms[o] = sqrt(e.r0 * e.r0 + e.r1 * e.r1);
}

static inline int stream_grid(int n) {
  const int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);  // memory-bound streams: cap at 8 blocks x 256 CUs, grid-stride the rest
}

// lanes per landmark of the landmark-major kernels (k_lm_lin, k_lm_backsub, k_lm_outliers): DevProblem::lm_group, chosen at upload from the map's mean
// track length (round 6) — the 5-agent map's tracks average 10 observations (16 lanes: 63 % of the lanes busy), configs[4]'s 4.1 (16 lanes: 25 %)
template <typename F> static void with_group(int g, F&& f) {
  if (g == 4) f(std::integral_constant<int, 4>()); else if (g == 8) f(std::integral_constant<int, 8>()); else f(std::integral_constant<int, 16>());
}
static int lm_blocks(const DevProblem& P) { const int g = P.lm_group == 4 || P.lm_group == 8 ? P.lm_group : 16; const int groups = kBuildThreads / g; return (P.L + groups - 1) / groups; }

// side != nullptr: the per-keyframe reduction (diagonal blocks, gradient, right-hand side: compute-heavy re-linearisation) runs on the
// side stream beside the pair pass (off-diagonal blocks: L2-bound record reads) — both follow the landmark pass only and write
// disjoint entries; whatever comes next on `st` follows both.
// first pass alone (per-observation records, per-landmark blocks, cost partials: touches nothing of the pose system) — the caller
// enqueues it ahead of the clearing of the fronts
void launch_lm_lin(const DevProblem& P, double mu, hipStream_t st, DevSignal sig) {
  if (P.L == 0) return;
  const int nblk = lm_blocks(P);
  with_group(P.lm_group, [&](auto G) { hipLaunchKernelGGL(k_lm_lin<decltype(G)::value>, dim3(nblk), dim3(kBuildThreads), 0, st, P, mu, sig); });
}
// the two passes that write the pose system, behind launch_lm_lin on `st`
void launch_lm_build(const DevProblem& P, double mu, hipStream_t st, hipEvent_t pose_system_cleared, hipStream_t side, hipEvent_t ev_lin, hipEvent_t ev_kf, CholAux* ax) {
  // (ordering through the context's device flags when the caller passes its CholAux, through HIP events otherwise: common.hpp)
  auto record = [&](hipEvent_t e, hipStream_t s) { if (ax) ax->record(e, s); else (void)hipEventRecord(e, s); };
  auto wait = [&](hipStream_t s, hipEvent_t e0, hipEvent_t e1 = nullptr) { if (ax) ax->wait(s, e0, e1); else { if (e0) (void)hipStreamWaitEvent(s, e0, 0); if (e1) (void)hipStreamWaitEvent(s, e1, 0); } };
  if (P.L == 0) { if (pose_system_cleared) wait(st, pose_system_cleared); return; }
  const int nblk = lm_blocks(P);
  const bool fork = side != nullptr && ev_lin != nullptr && ev_kf != nullptr && P.npairs > 0;
  hipStream_t s2 = fork ? side : st;
  // the side stream follows the landmark linearisation; the first writers of the pose system (both streams) follow the clearing of the fronts
  // (one gate for both on the side stream: the clearing ends long before the landmark pass)
  if (fork) { record(ev_lin, st); wait(s2, ev_lin, pose_system_cleared); if (pose_system_cleared) wait(st, pose_system_cleared); }
  else if (pose_system_cleared) wait(s2, pose_system_cleared);
  hipLaunchKernelGGL(k_cost_finish, dim3(1), dim3(256), 0, s2, P, nblk);
  hipLaunchKernelGGL(k_kf_reduce, dim3(P.K), dim3(64), 0, s2, P);
  if (P.npairs) {
    static const int xcd_order = getenv("COVGPU_PAIR_XCD") == nullptr || atoi(getenv("COVGPU_PAIR_XCD")) != 0;
    const int nblk = (P.npairs + kPairsPerWg - 1) / kPairsPerWg;
    const int nchunk8 = (nblk + 8 * kPairChunk - 1) / (8 * kPairChunk);   // rounds of eight chunks
    hipLaunchKernelGGL(k_pair_blocks, dim3(xcd_order ? nchunk8 * 8 * kPairChunk : nblk), dim3(kPairLanes * kPairsPerWg), 0, st, P, xcd_order);
  }
  // (fork: the caller joins the side stream back — it has more on it)
}
// upload: keyframe-major copies of the observation stream (DevProblem::kobs), the Z slot of every observation, and the covisible-pair
// entries turned from observation indices into Z slots
__global__ __launch_bounds__(256) void k_kobs_build(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.O) return;
  const int o = P.kf_obs_idx[t];
  P.kobs[3 * (size_t)t] = P.obs_u[o]; P.kobs[3 * (size_t)t + 1] = P.obs_v[o]; P.kobs[3 * (size_t)t + 2] = P.obs_sigma[o];
  P.kobs_lm[t] = P.obs_lm[o];
  P.obs_zpos[o] = t;
}
__global__ __launch_bounds__(256) void k_remap_idx(size_t n, int* __restrict__ a, int* __restrict__ b, const int* __restrict__ map) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) { a[q] = map[a[q]]; b[q] = map[b[q]]; }
}
void launch_kobs_build(const DevProblem& P, int* pair_oa, int* pair_ob, size_t nent, hipStream_t st) {
  if (P.O == 0) return;
  hipLaunchKernelGGL(k_kobs_build, dim3((P.O + 255) / 256), dim3(256), 0, st, P);
  if (nent > 0) hipLaunchKernelGGL(k_remap_idx, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, nent, pair_oa, pair_ob, (const int*)P.obs_zpos);
}
void launch_lm_backsub(const DevProblem& P, const double* dp, double* out_all, hipStream_t st) {
  if (P.L == 0) return;
  with_group(P.lm_group, [&](auto G) { hipLaunchKernelGGL(k_lm_backsub<decltype(G)::value>, dim3(lm_blocks(P)), dim3(kBuildThreads), 0, st, P, dp, out_all); });
}
void launch_obs_jvp(const DevProblem& P, const double* v_all, hipStream_t st) {
  if (P.O == 0) return;
  hipLaunchKernelGGL(k_obs_jvp, dim3(stream_grid(P.O)), dim3(256), 0, st, P, v_all);
}
void launch_obs_cost(const DevProblem& P, const double* pose, const double* lm, hipStream_t st) {
  if (P.O == 0) return;
  hipLaunchKernelGGL(k_obs_cost, dim3(stream_grid(P.O)), dim3(256), 0, st, P, pose, lm);
}
void launch_obs_linearize(const DevProblem& P, double* r, double* Jp, double* Jl, double* cost, hipStream_t st) {
  if (P.O == 0) return;
  hipLaunchKernelGGL(k_obs_linearize, dim3((P.O + 255) / 256), dim3(256), 0, st, P, r, Jp, Jl, cost);
}
// Map maintenance after the outlier round, on the device (optimization_be.cpp:270-290 + Map::Clean / RemoveLandmarkOutliers,
// map_be.cpp:448-454, 698-743): per observation "loss-corrected whitened residual norm > threshold -> erase", per landmark
// the number of observations it keeps; a landmark left with fewer than two is what RemoveLandmarkOutliers drops. One
// sub-wave group per landmark like the linearisation, the residual evaluated in place: only O bytes + L ints go back to
// the host instead of O doubles. counts[0] += erased observations, counts[1] += landmarks left with < 2.
template <int G>
__global__ __launch_bounds__(kBuildThreads) void k_lm_outliers(DevProblem P, double th, unsigned char* __restrict__ erase, int* __restrict__ left,
                                                               unsigned long long* __restrict__ counts) {
  constexpr int GROUPS = kBuildThreads / G;
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const int l = blockIdx.x * GROUPS + grp;
  const bool lm_ok = l < P.L;
  const int o0 = lm_ok ? P.lm_obs_ptr[l] : 0;
  const int nobs = lm_ok ? P.lm_obs_ptr[l + 1] - o0 : 0;
  double kept = 0.0, bad = 0.0;
  for (int a = lane; a < nobs; a += G) {
    ObsLin e;
    eval_obs<false>(P, P.pose, P.lm, o0 + a, P.obs_kf[o0 + a], l, e);
    const bool out = sqrt(e.r0 * e.r0 + e.r1 * e.r1) > th;
    erase[o0 + a] = out ? 1 : 0;
    kept += out ? 0.0 : 1.0; bad += out ? 1.0 : 0.0;
  }
  kept = group_sum<G>(kept); bad = group_sum<G>(bad);
  if (lm_ok && lane == 0) {
    left[l] = (int)kept;
    if (bad > 0.0) atomicAdd(&counts[0], (unsigned long long)bad);   // integer counters: order-independent
    if (kept < 2.0) atomicAdd(&counts[1], 1ull);
  }
}
void launch_lm_outliers(const DevProblem& P, double th, unsigned char* erase, int* left, unsigned long long* counts, hipStream_t st) {
  if (P.L == 0) return;
  with_group(P.lm_group, [&](auto G) { hipLaunchKernelGGL(k_lm_outliers<decltype(G)::value>, dim3(lm_blocks(P)), dim3(kBuildThreads), 0, st, P, th, erase, left, counts); });
}

void launch_obs_norms(const DevProblem& P, double* norms, hipStream_t st) {
  if (P.O == 0) return;
  hipLaunchKernelGGL(k_obs_norms, dim3((P.O + 255) / 256), dim3(256), 0, st, P, norms);
}

}  // namespace covgpu
