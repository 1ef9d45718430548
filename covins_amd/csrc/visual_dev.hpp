// visual_dev.hpp — device functions of the reprojection residual (R4 + R5 + R6) shared by the kernels of
// k_visual.hip (linearisation + landmark elimination) and k_tail.hip (the fused trust-region tail). Replaces
// robopt::reprojection::GlobalEuclideanReprError + aslam::PinholeCamera::project3 + ceres::CauchyLoss at
// optimization_be.cpp:487-525.
#pragma once
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

struct ObsLin {
  double r0, r1;
  double jp[12];  // 2x6 row-major: [dtheta(3), dp(3)]
  double jl[6];   // 2x3
  double cost;
};


// R4 + R6 for one observation. `fixed` zeroes the pose Jacobian (constant parameter block, opt_be.cpp:329-341).
template <bool JAC>
COV_DEV void eval_obs_uvs(const DevProblem& P, const double* __restrict__ pose, const double* __restrict__ lm, double mu_, double mv_, double sigma, int kf, int l,
                          ObsLin& out) {
  const int cam = P.kf_cam[kf];
  const double* ps = pose + 7 * kf;
  const double* ex = P.cam_extr + 7 * cam;
  const Q4 qws = ldq(ps), qsc = ldq(ex);
  const M3 Rws = qrot(qws), Rsc = qrot(qsc);
  const V3 lw = ld3(lm + 3 * l);
  const V3 ls = mulT(Rws, lw - ld3(ps + 4));
  const V3 lc = mulT(Rsc, ls - ld3(ex + 4));
  double u, v, jpi[6];
  const bool ok = project_point(lc, P.cam_intr + 4 * cam, P.cam_dist + 4 * cam, P.cam_dist_type[cam], u, v, JAC ? jpi : nullptr);
  if (!ok) {
    out.r0 = out.r1 = 0.0; out.cost = 0.0;
    if (JAC) {
#pragma unroll
      for (int k = 0; k < 12; ++k) out.jp[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) out.jl[k] = 0.0;
    }
    return;
  }
  const double is = 1.0 / sigma;
  double r0 = (u - mu_) * is, r1 = (v - mv_) * is;
  double c;
  const double sq = cauchy_scale(P.reproj_loss_a, r0 * r0 + r1 * r1, &c);
  out.r0 = r0 * sq; out.r1 = r1 * sq; out.cost = c;
  if (JAC) {
    const double w = is * sq;
    const bool fx = P.fixed[kf] != 0;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      // a = (w J_pi R_sc^T)_row  ->  a_c = w * sum_k jpi[row][k] Rsc[c][k]
      const V3 a = mul(Rsc, V3{jpi[3 * row] * w, jpi[3 * row + 1] * w, jpi[3 * row + 2] * w});
      const V3 jth = cross(a, ls);  // a^T [l_S]x
      const V3 jlw = mul(Rws, a);   // a^T R_ws^T
      out.jl[3 * row] = jlw.x; out.jl[3 * row + 1] = jlw.y; out.jl[3 * row + 2] = jlw.z;
      out.jp[6 * row] = fx ? 0.0 : jth.x; out.jp[6 * row + 1] = fx ? 0.0 : jth.y; out.jp[6 * row + 2] = fx ? 0.0 : jth.z;
      out.jp[6 * row + 3] = fx ? 0.0 : -jlw.x; out.jp[6 * row + 4] = fx ? 0.0 : -jlw.y; out.jp[6 * row + 5] = fx ? 0.0 : -jlw.z;
    }
  }
}

// the same from the landmark-major observation stream
template <bool JAC>
COV_DEV void eval_obs(const DevProblem& P, const double* __restrict__ pose, const double* __restrict__ lm, int o, int kf, int l, ObsLin& out) {
  eval_obs_uvs<JAC>(P, pose, lm, P.obs_u[o], P.obs_v[o], P.obs_sigma[o], kf, l, out);
}

}  // namespace covgpu
