// k_relpose.hip — batched Optimization::OptimizeRelativePose (optimization_be.cpp:620-831).
//
// The reference refines ONE relative pose T_AB per loop candidate from <= ~300 landmark correspondences with two
// reprojection residuals each (kNormal into camera A, kInverse into camera B), Cauchy(1), DOGLEG, 5 iterations, an outlier
// pass, 5 more iterations — a tiny, latency-bound problem, called serially from the place-recognition threads
// (placerec_be.cpp:116-165). Here many candidates are refined in ONE launch: one wavefront per keyframe pair, lanes over the
// correspondences, the 6x6 normal equations reduced with wave shuffles, the whole trust-region loop (Ceres 1.x dogleg
// defaults) on the device — no host round trip per iteration. Residual / Jacobian conventions are those of
// the GBA kernels (pose (+): q (x) Exp(dtheta), p + dp; projection: dev_math.hpp project_point; loss: Cauchy corrector).
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

struct RelBatch {
  int num;
  const int* ptr;
  const double *pB, *pA, *kpA, *kpB, *sigA, *sigB, *camA, *camB;
  const int *distA, *distB;
  double* T;               // [num][7] in/out
  unsigned char* outlier;  // [C] out
  int* inliers;            // [num] out
  double th;
  int min_inliers;
};

COV_DEV double wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// residual pair of one correspondence at (q, t); JAC: 4x6 Jacobian rows rA(2), rB(2), columns [dtheta, dp]; loss-corrected
template <bool JAC>
COV_DEV double rel_eval(const RelBatch& B, int b, int i, Q4 q, V3 t, double* r4, double* J) {
  const M3 R = qrot(q);
  const V3 PB = ld3(B.pB + 3 * (size_t)i), PA = ld3(B.pA + 3 * (size_t)i);
  double cost = 0.0;
  if (JAC) for (int k = 0; k < 24; ++k) J[k] = 0.0;
  {
    const V3 X = mul(R, PB) + t;
    double u, v, jpi[6];
    if (project_point(X, B.camA + 8 * b, B.camA + 8 * b + 4, B.distA[b], u, v, JAC ? jpi : nullptr)) {
      const double is = 1.0 / B.sigA[i];
      const double r0 = (u - B.kpA[2 * (size_t)i]) * is, r1 = (v - B.kpA[2 * (size_t)i + 1]) * is;
      double c; const double sq = cauchy_scale(1.0, r0 * r0 + r1 * r1, &c);
      r4[0] = sq * r0; r4[1] = sq * r1; cost += c;
      if (JAC) {
        const double w = is * sq;
        const M3 RS = mul(R, skew(PB));  // dX/dtheta = -R [P_B]x ; dX/dp = I
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const double a0 = jpi[3 * row] * w, a1 = jpi[3 * row + 1] * w, a2 = jpi[3 * row + 2] * w;
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) J[6 * row + c2] = -(a0 * RS.m[c2] + a1 * RS.m[3 + c2] + a2 * RS.m[6 + c2]);
          J[6 * row + 3] = a0; J[6 * row + 4] = a1; J[6 * row + 5] = a2;
        }
      }
    } else { r4[0] = 0.0; r4[1] = 0.0; }
  }
  {
    const V3 Y = mulT(R, PA - t);
    double u, v, jpi[6];
    if (project_point(Y, B.camB + 8 * b, B.camB + 8 * b + 4, B.distB[b], u, v, JAC ? jpi : nullptr)) {
      const double is = 1.0 / B.sigB[i];
      const double r0 = (u - B.kpB[2 * (size_t)i]) * is, r1 = (v - B.kpB[2 * (size_t)i + 1]) * is;
      double c; const double sq = cauchy_scale(1.0, r0 * r0 + r1 * r1, &c);
      r4[2] = sq * r0; r4[3] = sq * r1; cost += c;
      if (JAC) {
        const double w = is * sq;
        const M3 SY = skew(Y);           // dY/dtheta = [Y]x ; dY/dp = -R^T
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const double a0 = jpi[3 * row] * w, a1 = jpi[3 * row + 1] * w, a2 = jpi[3 * row + 2] * w;
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) {
            J[6 * (2 + row) + c2] = a0 * SY.m[c2] + a1 * SY.m[3 + c2] + a2 * SY.m[6 + c2];
            J[6 * (2 + row) + 3 + c2] = -(a0 * R.m[3 * c2] + a1 * R.m[3 * c2 + 1] + a2 * R.m[3 * c2 + 2]);  // (A R^T)[c2] = sum_k a_k R[c2][k]
          }
        }
      }
    } else { r4[2] = 0.0; r4[3] = 0.0; }
  }
  return cost;
}

COV_DEV bool chol6_solve(const double* A /*6x6 full*/, const double* b, double* x) {
  double L[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < j) d -= L[j][k] * L[j][k];
    if (!(d > 0.0)) return false;
    L[j][j] = sqrt(d);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i > j) {
        double s = A[6 * i + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) s -= L[i][k] * L[j][k];
        L[i][j] = s / L[j][j];
      }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { double s = b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < i) s -= L[i][k] * y[k];
    y[i] = s / L[i][i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { double s = y[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k > i) s -= L[k][i] * x[k];
    x[i] = s / L[i][i]; }
  return true;
}

// every lane holds identical copies of the 6-dof state and of the reduced 6x6 system: the trust-region decisions are taken
// redundantly (bit-identical: the shuffle reductions deliver the same sums to all lanes), no LDS, no barrier
COV_DEV void rel_dogleg(const RelBatch& B, int b, int o0, int n, int lane, Q4& q, V3& t, int iters) {
  double radius = 1e4, mu = 1e-8, alpha = 0.0, cost = 0.0;
  double H[36], g[6], D[6], gh[6], gn[6];
  bool reuse = false;
  auto linearise = [&]() {
    double h[21], gg[6], c = 0.0;
    for (int k = 0; k < 21; ++k) h[k] = 0.0;
    for (int k = 0; k < 6; ++k) gg[k] = 0.0;
    for (int a = lane; a < n; a += 64) {
      if (B.outlier[o0 + a]) continue;
      double r[4], J[24];
      c += rel_eval<true>(B, b, o0 + a, q, t, r, J);
      int e = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x) {
#pragma unroll
        for (int y = 0; y <= x; ++y, ++e) h[e] += J[x] * J[y] + J[6 + x] * J[6 + y] + J[12 + x] * J[12 + y] + J[18 + x] * J[18 + y];
        gg[x] += J[x] * r[0] + J[6 + x] * r[1] + J[12 + x] * r[2] + J[18 + x] * r[3];
      }
    }
    cost = wsum(c);
    int e = 0;
    for (int x = 0; x < 6; ++x)
      for (int y = 0; y <= x; ++y, ++e) { const double v = wsum(h[e]); H[6 * x + y] = v; H[6 * y + x] = v; }
    for (int x = 0; x < 6; ++x) { g[x] = wsum(gg[x]); D[x] = clamp_diag(H[7 * x]); }
  };
  auto total_cost = [&](Q4 qq, V3 tt) {
    double c = 0.0;
    for (int a = lane; a < n; a += 64) { if (B.outlier[o0 + a]) continue; double r[4]; c += rel_eval<false>(B, b, o0 + a, qq, tt, r, nullptr); }
    return wsum(c);
  };
  linearise();
  for (int it = 0; it < iters; ++it) {
    double gmax = 0.0;
    for (int k = 0; k < 6; ++k) gmax = fmax(gmax, fabs(g[k]));
    if (gmax <= 1e-10) break;
    bool ok = true;
    if (!reuse) {
      double g2 = 0.0, v[6], qf = 0.0;
      for (int k = 0; k < 6; ++k) { gh[k] = g[k] / D[k]; v[k] = gh[k] / D[k]; g2 += gh[k] * gh[k]; }
      for (int x = 0; x < 6; ++x) { double s = 0.0; for (int y = 0; y < 6; ++y) s += H[6 * x + y] * v[y]; qf += v[x] * s; }
      alpha = g2 / qf;
      ok = false;
      while (mu < 1.0) {
        double S[36], nb[6];
        for (int k = 0; k < 36; ++k) S[k] = H[k];
        for (int k = 0; k < 6; ++k) { if (H[7 * k] == 0.0) S[7 * k] = 1.0; else S[7 * k] += mu * D[k] * D[k]; nb[k] = -g[k]; }
        if (chol6_solve(S, nb, gn)) { ok = true; break; }
        mu *= 10.0;
      }
    }
    double step[6], step_norm = 0.0, model = 0.0;
    if (ok) {
      double gn2 = 0.0, g2 = 0.0, gdot = 0.0;
      for (int k = 0; k < 6; ++k) { const double a = D[k] * gn[k]; gn2 += a * a; g2 += gh[k] * gh[k]; gdot += gh[k] * a; }
      const double gn_norm = sqrt(gn2), g_norm = sqrt(g2);
      double cg, cn;
      if (gn_norm <= radius) { cg = 0.0; cn = 1.0; step_norm = gn_norm; }
      else if (g_norm * alpha >= radius) { cg = -radius / g_norm; cn = 0.0; step_norm = radius; }
      else {
        const double b_dot_a = -alpha * gdot, a_sq = alpha * alpha * g2;
        const double bma = gn2 - 2.0 * b_dot_a + a_sq, c = b_dot_a - a_sq;
        const double dd = sqrt(c * c + bma * (radius * radius - a_sq));
        const double beta = (c <= 0.0) ? (dd - c) / bma : (radius * radius - a_sq) / (dd + c);
        cg = -alpha * (1.0 - beta); cn = beta; step_norm = radius;
      }
      double gs = 0.0, sHs = 0.0;
      for (int k = 0; k < 6; ++k) step[k] = cg * gh[k] / D[k] + cn * gn[k];
      for (int x = 0; x < 6; ++x) { double s = 0.0; for (int y = 0; y < 6; ++y) s += H[6 * x + y] * step[y]; sHs += step[x] * s; gs += g[x] * step[x]; }
      model = -(gs + 0.5 * sHs);
    }
    if (!ok || !(model > 0.0)) { mu *= 10.0; reuse = false; if (mu >= 1.0 && !ok) break; continue; }
    const Q4 qn = qnormalize(qmul(q, qexp(V3{step[0], step[1], step[2]})));
    const V3 tn = V3{t.x + step[3], t.y + step[4], t.z + step[5]};
    const double cost_new = total_cost(qn, tn);
    const double rho = (cost - cost_new) / model;
    if (rho > 1e-3) {
      const bool conv = fabs(cost - cost_new) <= 1e-6 * cost;
      q = qn; t = tn;
      if (rho < 0.25) radius *= 0.5;
      if (rho > 0.75) radius = fmax(radius, 3.0 * step_norm);
      mu = fmax(1e-8, 2.0 * mu / 10.0);
      reuse = false;
      linearise();
      if (conv) break;
    } else { radius *= 0.5; reuse = true; }
  }
}

__global__ __launch_bounds__(64) void k_relpose(RelBatch B) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int o0 = B.ptr[b], n = B.ptr[b + 1] - o0;
  Q4 q = qnormalize(ldq(B.T + 7 * (size_t)b));
  V3 t = ld3(B.T + 7 * (size_t)b + 4);
  for (int a = lane; a < n; a += 64) B.outlier[o0 + a] = 0;
  rel_dogleg(B, b, o0, n, lane, q, t, 5);                       // max_num_iterations = 5 (optimization_be.cpp:787)
  double bad = 0.0;
  for (int a = lane; a < n; a += 64) {                          // problem.Evaluate + threshold (:791-811)
    double r[4];
    rel_eval<false>(B, b, o0 + a, q, t, r, nullptr);
    const bool out = sqrt(r[0] * r[0] + r[1] * r[1]) > B.th || sqrt(r[2] * r[2] + r[3] * r[3]) > B.th;
    B.outlier[o0 + a] = out ? 1 : 0;
    bad += out ? 1.0 : 0.0;
  }
  const int nbad = (int)wsum(bad);
  if (n - nbad < B.min_inliers) { if (lane == 0) B.inliers[b] = 0; return; }   // :813-815, T12 untouched
  rel_dogleg(B, b, o0, n, lane, q, t, 5);                       // :817-818
  if (lane == 0) {
    double* T = B.T + 7 * (size_t)b;
    T[0] = q.x; T[1] = q.y; T[2] = q.z; T[3] = q.w; T[4] = t.x; T[5] = t.y; T[6] = t.z;
    B.inliers[b] = n - nbad;
  }
}

void launch_relpose(int num, const int* ptr, const double* pB, const double* pA, const double* kpA, const double* kpB, const double* sigA,
                    const double* sigB, const double* camA, const int* distA, const double* camB, const int* distB, double th, int min_inliers,
                    double* T, unsigned char* outlier, int* inliers, hipStream_t st) {
  if (num <= 0) return;
  RelBatch B{num, ptr, pB, pA, kpA, kpB, sigA, sigB, camA, camB, distA, distB, T, outlier, inliers, th, min_inliers};
  hipLaunchKernelGGL(k_relpose, dim3(num), dim3(64), 0, st, B);
}

}  // namespace covgpu
