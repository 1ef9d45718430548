// covo_relpose.cpp — CPU oracle for Optimization::OptimizeRelativePose (optimization_be.cpp:620-831).
//
// TEST INFRASTRUCTURE ONLY (see covo_solver.cpp). PARITY UNPINNED: robopt::reprojection::RelativeEuclideanReprError and
// Ceres are not vendored in the reference; the residual below is the restatement its call site implies.
//
// One 6-dof unknown T_AB (pose block [q, p], R1 (+)), two reprojection residuals per correspondence i (:643-:777):
//   kNormal   r_A = ( pi_A( R_AB P_B + t_AB )      - kp_A ) / sigma_A      P_B = landmark of kf2 in camera-B frame  (:655-656)
//   kInverse  r_B = ( pi_B( R_AB^T (P_A - t_AB) )  - kp_B ) / sigma_B      P_A = landmark of kf1 in camera-A frame  (:653-654)
// sigma = (octave + 1) * 2 (:658, :717), Cauchy(1) on every block (:623-624, :771-776). Solve: DOGLEG, 5 iterations (:783-789);
// correspondences with |r_A| or |r_B| (loss-corrected, as problem.Evaluate returns them) above th_outlier_align are removed
// (:791-811); fewer than 12 left -> return 0 with T12 untouched (:813-815); else 5 more iterations (:817-822).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../include/covgpu.h"
#include "covo_residuals.hpp"

namespace covo {

struct RelProblem {
  int n;
  const double *pB, *pA, *kpA, *kpB, *sigA, *sigB, *camA, *camB;  // cam = fx fy cx cy d0 d1 d2 d3
  int distA, distB;
  std::vector<char> off;  // removed correspondences
};

// residual pair of correspondence i at T, loss-corrected; J rows: rA(2) then rB(2), columns [dtheta(3), dp(3)]
static void rel_eval(const RelProblem& P, const Pose& T, int i, double* r4, Mat<4, 6>* J, double* cost) {
  const Mat3 R = T.q.R();
  const Vec3 PB = v3(P.pB + 3 * i), PA = v3(P.pA + 3 * i);
  *cost = 0;
  if (J) *J = Mat<4, 6>();
  {  // kNormal, camera A
    const Vec3 X = R * PB + T.p;
    double uv[2]; Mat<2, 3> Jpi;
    if (project(X, P.camA, P.camA + 4, P.distA, uv, J ? &Jpi : nullptr)) {
      const double is = 1.0 / P.sigA[i];
      double r0 = (uv[0] - P.kpA[2 * i]) * is, r1 = (uv[1] - P.kpA[2 * i + 1]) * is, c;
      const double sq = cauchy(1.0, r0 * r0 + r1 * r1, &c);
      r4[0] = sq * r0; r4[1] = sq * r1; *cost += c;
      if (J) {
        const Mat<2, 3> A = Jpi * (is * sq);
        const Mat<2, 3> Jth = A * (R * skew(PB)) * -1.0;   // dX/dtheta = -R [P_B]x
        J->set(0, 0, Jth); J->set(0, 3, A);                // dX/dp = I
      }
    } else { r4[0] = r4[1] = 0; }
  }
  {  // kInverse, camera B
    const Vec3 Y = R.T() * (PA - T.p);
    double uv[2]; Mat<2, 3> Jpi;
    if (project(Y, P.camB, P.camB + 4, P.distB, uv, J ? &Jpi : nullptr)) {
      const double is = 1.0 / P.sigB[i];
      double r0 = (uv[0] - P.kpB[2 * i]) * is, r1 = (uv[1] - P.kpB[2 * i + 1]) * is, c;
      const double sq = cauchy(1.0, r0 * r0 + r1 * r1, &c);
      r4[2] = sq * r0; r4[3] = sq * r1; *cost += c;
      if (J) {
        const Mat<2, 3> A = Jpi * (is * sq);
        J->set(2, 0, A * skew(Y));                         // dY/dtheta = [Y]x
        J->set(2, 3, (A * R.T()) * -1.0);                  // dY/dp = -R^T
      }
    } else { r4[2] = r4[3] = 0; }
  }
}

static double rel_cost(const RelProblem& P, const Pose& T) {
  double c = 0;
  for (int i = 0; i < P.n; ++i) { if (P.off[i]) continue; double r[4], ci; rel_eval(P, T, i, r, nullptr, &ci); c += ci; }
  return c;
}

static bool chol6_solve(Mat<6, 6> A, Mat<6, 1> b, Mat<6, 1>* x) {
  double L[6][6] = {};
  for (int j = 0; j < 6; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    if (!(d > 0)) return false;
    L[j][j] = std::sqrt(d);
    for (int i = j + 1; i < 6; ++i) { double s = A(i, j); for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k]; L[i][j] = s / L[j][j]; }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * (*x)[k]; (*x)[i] = s / L[i][i]; }
  return true;
}

static inline double clampd6(double h) { return std::min(std::max(std::sqrt(std::max(h, 0.0)), 1e-6), 1e32); }

// traditional dogleg on the 6-dof problem, Ceres 1.x defaults, `iters` iterations incl. rejected ones
static void rel_dogleg(const RelProblem& P, Pose& T, int iters) {
  double radius = 1e4, mu = 1e-8;
  Mat<6, 6> H; Mat<6, 1> g, gn, gh; double cost = 0, alpha = 0, D[6];
  bool reuse = false;
  auto linearise = [&]() {
    H = Mat<6, 6>(); g = Mat<6, 1>(); cost = 0;
    for (int i = 0; i < P.n; ++i) {
      if (P.off[i]) continue;
      double r[4], c; Mat<4, 6> J;
      rel_eval(P, T, i, r, &J, &c);
      cost += c;
      Mat<4, 1> rv; for (int k = 0; k < 4; ++k) rv[k] = r[k];
      H += J.T() * J; g += J.T() * rv;
    }
    for (int k = 0; k < 6; ++k) D[k] = clampd6(H(k, k));
  };
  linearise();
  for (int it = 0; it < iters; ++it) {
    double gmax = 0; for (int k = 0; k < 6; ++k) gmax = std::max(gmax, std::fabs(g[k]));
    if (gmax <= 1e-10) break;
    bool ok = true;
    if (!reuse) {
      double gg = 0; Mat<6, 1> v;
      for (int k = 0; k < 6; ++k) { gh[k] = g[k] / D[k]; v[k] = gh[k] / D[k]; gg += gh[k] * gh[k]; }
      const Mat<6, 1> Hv = H * v;
      double q = 0; for (int k = 0; k < 6; ++k) q += v[k] * Hv[k];
      alpha = gg / q;
      ok = false;
      while (mu < 1.0) {
        Mat<6, 6> S = H;
        for (int k = 0; k < 6; ++k) { if (H(k, k) == 0.0) S(k, k) = 1.0; else S(k, k) += mu * D[k] * D[k]; }
        if (chol6_solve(S, g * -1.0, &gn)) { ok = true; break; }
        mu *= 10.0;
      }
    }
    Mat<6, 1> step; double step_norm = 0, model = 0;
    if (ok) {
      double gn2 = 0, g2 = 0, gdot = 0;
      for (int k = 0; k < 6; ++k) { const double a = D[k] * gn[k]; gn2 += a * a; g2 += gh[k] * gh[k]; gdot += gh[k] * a; }
      const double gn_norm = std::sqrt(gn2), g_norm = std::sqrt(g2);
      double cg, cn;
      if (gn_norm <= radius) { cg = 0; cn = 1; step_norm = gn_norm; }
      else if (g_norm * alpha >= radius) { cg = -radius / g_norm; cn = 0; step_norm = radius; }
      else {
        const double b_dot_a = -alpha * gdot, a_sq = alpha * alpha * g2;
        const double bma = gn2 - 2 * b_dot_a + a_sq, c = b_dot_a - a_sq;
        const double dd = std::sqrt(c * c + bma * (radius * radius - a_sq));
        const double beta = (c <= 0) ? (dd - c) / bma : (radius * radius - a_sq) / (dd + c);
        cg = -alpha * (1 - beta); cn = beta; step_norm = radius;
      }
      for (int k = 0; k < 6; ++k) step[k] = cg * gh[k] / D[k] + cn * gn[k];
      const Mat<6, 1> Hs = H * step;
      double gs = 0, sHs = 0; for (int k = 0; k < 6; ++k) { gs += g[k] * step[k]; sHs += step[k] * Hs[k]; }
      model = -(gs + 0.5 * sHs);
    }
    if (!ok || !(model > 0.0)) { mu *= 10.0; reuse = false; if (mu >= 1.0 && !ok) break; continue; }
    const Pose Tn = pose_plus(T, step.a);
    const double cost_new = rel_cost(P, Tn);
    const double rho = (cost - cost_new) / model;
    if (rho > 1e-3) {
      const bool conv = std::fabs(cost - cost_new) <= 1e-6 * cost;
      T = Tn;
      if (rho < 0.25) radius *= 0.5;
      if (rho > 0.75) radius = std::max(radius, 3.0 * step_norm);
      mu = std::max(1e-8, 2.0 * mu / 10.0);
      reuse = false;
      linearise();
      if (conv) break;
    } else { radius *= 0.5; reuse = true; }
  }
}

}  // namespace covo

using namespace covo;

// One problem. Returns the number of inliers (0: fewer than `min_inliers` survived, T_ab left untouched).
extern "C" int covo_relpose(int n, const double* pB, const double* pA, const double* kpA, const double* kpB, const double* sigA, const double* sigB,
                            const double* camA, int distA, const double* camB, int distB, double th_outlier, int min_inliers, double* T_ab,
                            unsigned char* outlier) {
  RelProblem P{n, pB, pA, kpA, kpB, sigA, sigB, camA, camB, distA, distB, std::vector<char>(n, 0)};
  Pose T(T_ab);
  T.q = T.q.normalized();
  rel_dogleg(P, T, 5);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    double r[4], c; rel_eval(P, T, i, r, nullptr, &c);
    const bool out = std::sqrt(r[0] * r[0] + r[1] * r[1]) > th_outlier || std::sqrt(r[2] * r[2] + r[3] * r[3]) > th_outlier;
    outlier[i] = out; P.off[i] = out; bad += out;
  }
  if (n - bad < min_inliers) return 0;
  rel_dogleg(P, T, 5);
  T.store(T_ab);
  return n - bad;
}

// raw residual pair + Jacobian (finite-difference test of the oracle itself)
extern "C" void covo_relpose_residual(const double* T_ab, const double* pB, const double* pA, const double* kpA, const double* kpB, double sigA, double sigB,
                                      const double* camA, int distA, const double* camB, int distB, double* r4, double* J24) {
  RelProblem P{1, pB, pA, kpA, kpB, &sigA, &sigB, camA, camB, distA, distB, std::vector<char>(1, 0)};
  Mat<4, 6> J; double c;
  rel_eval(P, Pose(T_ab), 0, r4, J24 ? &J : nullptr, &c);
  if (J24) for (int k = 0; k < 24; ++k) J24[k] = J[k];
}
