// between_dev.hpp — device function of the SE3 between factor (R7, SURVEY.md A.3: robopt::posegraph::SixDofBetweenError,
// optimization_be.cpp:252,554,934,968,1017) shared by k_between.hip and k_tail.hip (the fused trust-region tail).
#pragma once
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

// r[6], J[6x12] = [J_i | J_j] (whitened by sqrt_info, loss-corrected, fixed poses zeroed); returns cost.
template <bool JAC>
COV_DEV double eval_edge(const DevProblem& P, const double* __restrict__ pose, int e, double* r, double* J) {
  const int i = P.edge_i[e], j = P.edge_j[e];
  const double *Ti = pose + 7 * i, *Tj = pose + 7 * j, *Tm = P.edge_meas + 7 * (size_t)e;
  const double* S = P.edge_sqrt_info + 36 * (size_t)e;
  const Q4 qi = ldq(Ti), qj = ldq(Tj), qm = ldq(Tm);
  const M3 Ri = qrot(qi);
  const V3 that = mulT(Ri, ld3(Tj + 4) - ld3(Ti + 4));
  const Q4 er = qmul(qconj(qm), qmul(qconj(qi), qj));
  const double u[6] = {2.0 * er.x, 2.0 * er.y, 2.0 * er.z, that.x - Tm[4], that.y - Tm[5], that.z - Tm[6]};
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t += S[6 * a + k] * u[k];
    r[a] = t; s += t * t;
  }
  double cost;
  const double sq = cauchy_scale(P.edge_loss_a[e], s, &cost);
#pragma unroll
  for (int a = 0; a < 6; ++a) r[a] *= sq;
  if (JAC) {
    // un-whitened 6x12: rows [rot(3); trans(3)], cols [dth_i dp_i dth_j dp_j]
    double A[72];
#pragma unroll
    for (int k = 0; k < 72; ++k) A[k] = 0.0;
    const bool fi = P.fixed[i] != 0, fj = P.fixed[j] != 0;
    const M3 RiT = transpose(Ri);
    if (!fi) {
      const M3 a = mul(quat_lr3(er, -1.0), transpose(qrot(qm)));  // R3(e) R_m^T
      const M3 sk = skew(that);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          A[12 * rr + c] = -a.m[3 * rr + c];
          A[12 * (3 + rr) + c] = sk.m[3 * rr + c];
          A[12 * (3 + rr) + 3 + c] = -RiT.m[3 * rr + c];
        }
    }
    if (!fj) {
      const M3 l = quat_lr3(er, 1.0);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          A[12 * rr + 6 + c] = l.m[3 * rr + c];
          A[12 * (3 + rr) + 9 + c] = RiT.m[3 * rr + c];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) t += S[6 * a + k] * A[12 * k + c];
        J[12 * a + c] = t * sq;
      }
  }
  return cost;
}

}  // namespace covgpu
