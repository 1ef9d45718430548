// k_between.hip — SE3 between factors (loop closures in GBA; loop + odometry edges in PGO) on gfx950.
//
// Replaces R7 robopt::posegraph::SixDofBetweenError(q_12, t_12, sqrt_info, kImu), SizedCostFunction<6,7,7,7,7>,
// constructed at covins_backend/src/covins_backend/optimization_be.cpp:252,554 (GBA loop edges, sqrt_info =
// diag(100 I3, 1e4 I3), Cauchy(1) in round 2) and :934,968,1017 (PGO: loops, successor edges from VIO poses,
// five previous-neighbour edges). Extrinsics are ignored for kImu. Contract: SURVEY.md A.3 / A.5.
// Also the PGO tail (opt_be.cpp:1046-1047, 1066-1081): velocity rotation and landmark re-anchoring.
//
// One thread per edge: E is O(6 K) in PGO and O(10) in GBA; the 6x12 Jacobian stays in registers.
#include "common.hpp"
#include "dev_math.hpp"
#include "reduce.hpp"
#include "between_dev.hpp"

namespace covgpu {
using namespace covdev;

// per-edge normal-equation pieces, written with plain stores: Hii(36) Hjj(36) Hij(36) gi(6) gj(6) hdi(6) hdj(6)
constexpr int kEdgeRec = 132;
__global__ __launch_bounds__(64) void k_edge_build(DevProblem P) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (e < P.E) {
    double r[6], J[72];
    cost = eval_edge<true>(P, P.pose, e, r, J);
    double* o = P.edgeOut + (size_t)kEdgeRec * e;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double gi = 0.0, gj = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { gi += J[12 * k + a] * r[k]; gj += J[12 * k + 6 + a] * r[k]; }
      o[108 + a] = gi; o[114 + a] = gj;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double hii = 0.0, hjj = 0.0, hij = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          hii += J[12 * k + a] * J[12 * k + b]; hjj += J[12 * k + 6 + a] * J[12 * k + 6 + b]; hij += J[12 * k + a] * J[12 * k + 6 + b];
        }
        o[6 * a + b] = hii; o[36 + 6 * a + b] = hjj; o[72 + 6 * a + b] = hij;
        if (a == b) { o[120 + a] = hii; o[126 + a] = hjj; }
      }
    }
  }
  cost = wave_sum(cost);
  part_put(P, SC_COST, P.part_edge + blockIdx.x, cost);
}

// fixed-order sums of the per-edge pieces: (a) per keyframe: diagonal block of C, gradient, rhs, diag(J^T J);
// (b) per unique keyframe pair: off-diagonal block of C. One thread per destination entry.
__global__ __launch_bounds__(256) void k_edge_gather_kf(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int kf = t / 48, q = t - 48 * kf;
  if (kf >= P.K) return;
  double acc = 0.0;
  for (int s2 = P.kf_edge_ptr[kf]; s2 < P.kf_edge_ptr[kf + 1]; ++s2) {
    const int ent = P.kf_edge_ent[s2], e = ent >> 1, role = ent & 1;
    const double* o = P.edgeOut + (size_t)kEdgeRec * e;
    acc += (q < 36) ? o[36 * role + q] : (q < 42 ? o[108 + 6 * role + (q - 36)] : o[120 + 6 * role + (q - 42)]);
  }
  if (P.kf_edge_ptr[kf] == P.kf_edge_ptr[kf + 1]) return;
  const int pos = P.perm[kf];
  if (q < 36) {
    const int r = q / 6, c = q - 6 * r;
    if (c <= r) *c_entry(P, pos, pos, r, c) += acc;
  } else if (q < 42) {
    P.grad[(size_t)P.D * kf + q - 36] += acc; P.bred[(size_t)P.D * kf + q - 36] -= acc;
  } else {
    P.hdiag[(size_t)P.D * kf + q - 42] += acc;
  }
}
__global__ __launch_bounds__(256) void k_edge_gather_pair(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int pr = t / 36, q = t - 36 * pr;
  if (pr >= P.nepairs) return;
  const int r = q / 6, c = q - 6 * r;
  double acc = 0.0;
  for (int s2 = P.epair_ptr[pr]; s2 < P.epair_ptr[pr + 1]; ++s2) {
    const int ent = P.epair_ent[s2], e = ent >> 1, tr = ent & 1;
    const double* h = P.edgeOut + (size_t)kEdgeRec * e + 72;   // Hij: rows = dims of edge_i, cols = dims of edge_j
    acc += tr ? h[6 * c + r] : h[6 * r + c];
  }
  *c_entry(P, P.epair_i[pr], P.epair_j[pr], r, c) += acc;
}

__global__ __launch_bounds__(64) void k_edge_jvp(DevProblem P, const double* __restrict__ v_all) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (e < P.E) {
    double r[6], J[72];
    eval_edge<true>(P, P.pose, e, r, J);
    const double* vi = v_all + (size_t)P.D * P.edge_i[e];
    const double* vj = v_all + (size_t)P.D * P.edge_j[e];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) t += J[12 * a + c] * vi[c] + J[12 * a + 6 + c] * vj[c];
      acc += t * t;
    }
  }
  acc = wave_sum(acc);
  part_put(P, SC_JV2, P.part_edge + blockIdx.x, acc);
}

__global__ __launch_bounds__(64) void k_edge_cost(DevProblem P, const double* __restrict__ pose) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (e < P.E) {
    double r[6];
    acc = eval_edge<false>(P, pose, e, r, nullptr);
  }
  acc = wave_sum(acc);
  part_put(P, SC_COST, P.part_edge + blockIdx.x, acc);
}

__global__ __launch_bounds__(64) void k_edge_linearize(DevProblem P, double* r_out, double* J_out, double* cost) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.E) return;
  double r[6], J[72];
  cost[e] = eval_edge<true>(P, P.pose, e, r, J);
  for (int k = 0; k < 6; ++k) r_out[6 * (size_t)e + k] = r[k];
  for (int k = 0; k < 72; ++k) J_out[72 * (size_t)e + k] = J[k];
}

// PGO tail. One thread per keyframe (velocity) / landmark (re-anchoring); both embarrassingly parallel.
__global__ __launch_bounds__(256) void k_reanchor(int K, const double* __restrict__ po, const double* __restrict__ pn, double* vel, int L,
                                                   const int* __restrict__ ref, double* lm) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (vel && t < K) {
    const M3 Rn = qrot(qnormalize(ldq(pn + 7 * t))), Ro = qrot(qnormalize(ldq(po + 7 * t)));
    const V3 v = mul(Rn, mulT(Ro, ld3(vel + 3 * t)));
    vel[3 * t] = v.x; vel[3 * t + 1] = v.y; vel[3 * t + 2] = v.z;
  }
  if (t < L) {
    const int k = ref[t];
    if (k >= 0) {
      const M3 Ro = qrot(qnormalize(ldq(po + 7 * k))), Rn = qrot(qnormalize(ldq(pn + 7 * k)));
      const V3 ps = mulT(Ro, ld3(lm + 3 * t) - ld3(po + 7 * k + 4));
      const V3 pw = mul(Rn, ps) + ld3(pn + 7 * k + 4);
      lm[3 * t] = pw.x; lm[3 * t + 1] = pw.y; lm[3 * t + 2] = pw.z;
    }
  }
}

void launch_edge_build(const DevProblem& P, hipStream_t st) {
  if (P.E == 0) return;
  hipLaunchKernelGGL(k_edge_build, dim3((P.E + 63) / 64), dim3(64), 0, st, P);
}
void launch_edge_gather(const DevProblem& P, hipStream_t st) {
  if (P.E == 0) return;
  hipLaunchKernelGGL(k_edge_gather_kf, dim3((48 * P.K + 255) / 256), dim3(256), 0, st, P);
  if (P.nepairs) hipLaunchKernelGGL(k_edge_gather_pair, dim3((36 * P.nepairs + 255) / 256), dim3(256), 0, st, P);
}
void launch_edge_jvp(const DevProblem& P, const double* v_all, hipStream_t st) {
  if (P.E == 0) return;
  hipLaunchKernelGGL(k_edge_jvp, dim3((P.E + 63) / 64), dim3(64), 0, st, P, v_all);
}
void launch_edge_cost(const DevProblem& P, const double* pose, hipStream_t st) {
  if (P.E == 0) return;
  hipLaunchKernelGGL(k_edge_cost, dim3((P.E + 63) / 64), dim3(64), 0, st, P, pose);
}
void launch_edge_linearize(const DevProblem& P, double* r, double* J, double* cost, hipStream_t st) {
  if (P.E == 0) return;
  hipLaunchKernelGGL(k_edge_linearize, dim3((P.E + 63) / 64), dim3(64), 0, st, P, r, J, cost);
}
void launch_reanchor(int K, const double* pose_old, const double* pose_new, double* vel, int L, const int* ref, double* lm,
                     hipStream_t st) {
  const int n = K > L ? K : L;
  if (n == 0) return;
  hipLaunchKernelGGL(k_reanchor, dim3((n + 255) / 256), dim3(256), 0, st, K, pose_old, pose_new, vel, L, ref, lm);
}

}  // namespace covgpu
