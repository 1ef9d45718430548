// covo_residuals.hpp — CPU restatement of the residual blocks on the COVINS GBA / PGO path.
//
// TEST INFRASTRUCTURE ONLY: the oracle is the checker for the HIP path, never the product.
//
// PARITY UNPINNED: the arithmetic of these blocks lives in robopt_open@fix_imu_residual, aslam_cv2 and
// Ceres 1.x (dependencies.rosinstall:40-41,64-70), none of which is vendored in /root/reference, and the
// reference ships no tests or golden vectors for this path. What is restated here is
// the published algorithm of those dependencies as written out in SURVEY.md Appendix A, anchored on the
// reference's own call sites (cited per function). The oracle's self-consistency (analytic Jacobians vs
// central differences, closed-form preintegration cases, dense-vs-Schur equality) is pinned in tests/.
#pragma once
#include "covo_math.hpp"

namespace covo {

struct Pose {
  Quat q;
  Vec3 p;
  Pose() {}
  explicit Pose(const double* a) : q(a), p(v3(a + 4)) {}
  void store(double* a) const { a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w; a[4] = p[0]; a[5] = p[1]; a[6] = p[2]; }
};

// R1 — pose (+): q+ = q (x) Exp(dtheta), p+ = p + dp; tangent order [dtheta, dp].
// Replaces robopt::local_param::PoseQuaternionLocalParameterization (opt_be.cpp:303,328).
// Ceres2Transform re-normalises the quaternion (utils_base.cpp:38-40); we do so on every (+).
inline Pose pose_plus(const Pose& x, const double* d6) {
  Pose y;
  y.q = (x.q * quat_exp(v3(d6))).normalized();
  y.p = x.p + v3(d6 + 3);
  return y;
}

// R6 — ceres::CauchyLoss(a) with the Ceres 1.x corrector for rho'' < 0: scale r and J by sqrt(rho').
// (opt_be.cpp:302,523,555,840-841). Returns sqrt(rho'), writes cost = rho/2.
inline double cauchy(double a, double s, double* cost) {
  if (a <= 0.0) { *cost = 0.5 * s; return 1.0; }
  const double b = a * a, t = 1.0 + s / b;
  *cost = 0.5 * b * std::log(t);
  return std::sqrt(1.0 / t);
}

// R5 — aslam PinholeCamera + RadTan / Equidistant distortion: normalised point -> pixel and d(pixel)/d(l_C).
// (EuRoC case orb_slam3/src/KeyFrame.cc:64-65; dispatch opt_be.cpp:483-521).
inline bool project(const Vec3& lc, const double* intr, const double* dist, int dist_type, double* uv, Mat<2, 3>* Jpi) {
  const double X = lc[0], Y = lc[1], Z = lc[2];
  if (!(Z > 1e-10)) return false;  // behind the camera: residual and Jacobians are zeroed (A.2)
  const double iz = 1.0 / Z, x = X * iz, y = Y * iz;
  const double r2 = x * x + y * y;
  double xd, yd, dxx, dxy, dyx, dyy;
  if (dist_type == 0) {
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const double rad = k1 * r2 + k2 * r2 * r2, dr = k1 + 2 * k2 * r2;
    xd = x + x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    yd = y + y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y);
    dxx = 1 + rad + 2 * x * x * dr + 2 * p1 * y + 6 * p2 * x;
    dxy = 2 * x * y * dr + 2 * p1 * x + 2 * p2 * y;
    dyx = 2 * x * y * dr + 2 * p2 * y + 2 * p1 * x;
    dyy = 1 + rad + 2 * y * y * dr + 2 * p2 * x + 6 * p1 * y;
  } else {
    const double rho = std::sqrt(r2);
    if (rho < 1e-8) {
      xd = x; yd = y; dxx = 1; dxy = 0; dyx = 0; dyy = 1;
    } else {
      const double th = std::atan(rho), t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      const double thd = th * (1 + dist[0] * t2 + dist[1] * t4 + dist[2] * t6 + dist[3] * t8);
      const double dthd = 1 + 3 * dist[0] * t2 + 5 * dist[1] * t4 + 7 * dist[2] * t6 + 9 * dist[3] * t8;
      const double sc = thd / rho;
      const double dsc = (dthd / (1 + r2) * rho - thd) / r2;  // d(sc)/d(rho)
      xd = sc * x; yd = sc * y;
      dxx = sc + x * dsc * x / rho; dxy = x * dsc * y / rho;
      dyx = y * dsc * x / rho;      dyy = sc + y * dsc * y / rho;
    }
  }
  uv[0] = intr[0] * xd + intr[2];
  uv[1] = intr[1] * yd + intr[3];
  if (Jpi) {
    // d(xd,yd)/d(x,y) * d(x,y)/d(X,Y,Z), rows scaled by fx, fy
    const double a00 = iz, a02 = -x * iz, a11 = iz, a12 = -y * iz;
    (*Jpi)(0, 0) = intr[0] * (dxx * a00);
    (*Jpi)(0, 1) = intr[0] * (dxy * a11);
    (*Jpi)(0, 2) = intr[0] * (dxx * a02 + dxy * a12);
    (*Jpi)(1, 0) = intr[1] * (dyx * a00);
    (*Jpi)(1, 1) = intr[1] * (dyy * a11);
    (*Jpi)(1, 2) = intr[1] * (dyx * a02 + dyy * a12);
  }
  return true;
}

// R4 — robopt::reprojection::GlobalEuclideanReprError (opt_be.cpp:487-525).
// Un-corrected whitened residual and Jacobians; the caller applies the loss.
inline bool reproj(const Pose& Tws, const Pose& Tsc, const Vec3& lw, const double* intr, const double* dist, int dist_type,
                   const double* kp, double sigma, double* r, Mat<2, 6>* Jp, Mat<2, 3>* Jl) {
  const Mat3 Rws = Tws.q.R(), Rsc = Tsc.q.R();
  const Vec3 ls = Rws.T() * (lw - Tws.p);
  const Vec3 lc = Rsc.T() * (ls - Tsc.p);
  double uv[2];
  Mat<2, 3> Jpi;
  if (!project(lc, intr, dist, dist_type, uv, (Jp || Jl) ? &Jpi : nullptr)) {
    r[0] = r[1] = 0;
    if (Jp) *Jp = Mat<2, 6>();
    if (Jl) *Jl = Mat<2, 3>();
    return false;
  }
  const double is = 1.0 / sigma;
  r[0] = (uv[0] - kp[0]) * is;
  r[1] = (uv[1] - kp[1]) * is;
  if (Jp || Jl) {
    const Mat<2, 3> A = (Jpi * Rsc.T()) * is;  // (1/sigma) J_pi R_sc^T
    const Mat<2, 3> Jdth = A * skew(ls);
    const Mat<2, 3> Jlw = A * Rws.T();
    if (Jp) { Jp->set(0, 0, Jdth); Jp->set(0, 3, -Jlw); }
    if (Jl) *Jl = Jlw;
  }
  return true;
}

// R7 — robopt::posegraph::SixDofBetweenError, kImu (opt_be.cpp:554,934,968,1017).
// r = S [ 2 vec(q_m^-1 (x) q_1^-1 (x) q_2) ; R_1^T (p_2 - p_1) - t_m ], rotation rows first.
inline void between(const Pose& T1, const Pose& T2, const Pose& Tm, const Mat<6, 6>& S, Vec6* r, Mat<6, 6>* J1, Mat<6, 6>* J2) {
  const Mat3 R1 = T1.q.R();
  const Quat qhat = T1.q.inv() * T2.q;
  const Vec3 that = R1.T() * (T2.p - T1.p);
  const Quat e = Tm.q.inv() * qhat;
  Vec6 u;
  for (int k = 0; k < 3; ++k) { u[k] = 2.0 * e.vec()[k]; u[3 + k] = that[k] - Tm.p[k]; }
  *r = S * u;
  if (J1) {
    Mat<6, 6> A;
    A.set(0, 0, -(quat_R3(e) * Tm.q.R().T()));
    A.set(3, 0, skew(that));
    A.set(3, 3, -R1.T());
    *J1 = S * A;
  }
  if (J2) {
    Mat<6, 6> B;
    B.set(0, 0, quat_L3(e));
    B.set(3, 3, R1.T());
    *J2 = S * B;
  }
}

// R2 — robopt::imu::PreintegrationBase, VINS-Mono midpoint scheme.
// API seen at keyframe_be.cpp:187,195,203 and opt_be.cpp:396 (repropagate(ba,bg)).
struct Preint {
  Vec3 dp, dv;
  Quat dq;
  double dt = 0;
  Mat<15, 15> J, P;
  Vec3 ba, bg;  // linearisation biases
};
struct ImuNoise { double sa, sg, saw, sgw, g; };

inline void preintegrate(const double* first6, const double* samples7, int n, const Vec3& ba, const Vec3& bg,
                         const ImuNoise& nz, Preint* out) {
  Preint s;
  s.ba = ba; s.bg = bg;
  s.J = Mat<15, 15>::Identity();
  Vec3 a0 = v3(first6), w0 = v3(first6 + 3);
  double nd[18];
  for (int k = 0; k < 3; ++k) {
    nd[k] = nz.sa * nz.sa;       nd[3 + k] = nz.sg * nz.sg;
    nd[6 + k] = nz.sa * nz.sa;   nd[9 + k] = nz.sg * nz.sg;
    nd[12 + k] = nz.saw * nz.saw; nd[15 + k] = nz.sgw * nz.sgw;
  }
  const Mat3 I3 = Mat3::Identity();
  for (int i = 0; i < n; ++i) {
    const double dt = samples7[7 * i];
    const Vec3 a1 = v3(samples7 + 7 * i + 1), w1 = v3(samples7 + 7 * i + 4);
    const Vec3 w = (w0 + w1) * 0.5 - bg;
    const Quat dq1 = (s.dq * Quat(w[0] * dt * 0.5, w[1] * dt * 0.5, w[2] * dt * 0.5, 1.0)).normalized();
    const Mat3 Rq = s.dq.R(), Rr = dq1.R();
    const Vec3 abar = (Rq * (a0 - ba) + Rr * (a1 - ba)) * 0.5;
    const Vec3 dp1 = s.dp + s.dv * dt + abar * (0.5 * dt * dt);
    const Vec3 dv1 = s.dv + abar * dt;
    const Mat3 A0 = skew(a0 - ba), A1 = skew(a1 - ba), Om = skew(w);
    const Mat3 ImO = I3 - Om * dt;
    Mat<15, 15> F;
    F.set(0, 0, I3);
    F.set(0, 3, (Rq * A0) * (-0.25 * dt * dt) + (Rr * A1 * ImO) * (-0.25 * dt * dt));
    F.set(0, 6, I3 * dt);
    F.set(0, 9, (Rq + Rr) * (-0.25 * dt * dt));
    F.set(0, 12, (Rr * A1) * (0.25 * dt * dt * dt));
    F.set(3, 3, ImO);
    F.set(3, 12, I3 * (-dt));
    F.set(6, 3, (Rq * A0) * (-0.5 * dt) + (Rr * A1 * ImO) * (-0.5 * dt));
    F.set(6, 6, I3);
    F.set(6, 9, (Rq + Rr) * (-0.5 * dt));
    F.set(6, 12, (Rr * A1) * (0.5 * dt * dt));
    F.set(9, 9, I3);
    F.set(12, 12, I3);
    Mat<15, 18> V;
    V.set(0, 0, Rq * (0.25 * dt * dt));
    V.set(0, 3, (Rr * A1) * (-0.125 * dt * dt * dt));
    V.set(0, 6, Rr * (0.25 * dt * dt));
    V.set(0, 9, (Rr * A1) * (-0.125 * dt * dt * dt));
    V.set(3, 3, I3 * (0.5 * dt));
    V.set(3, 9, I3 * (0.5 * dt));
    V.set(6, 0, Rq * (0.5 * dt));
    V.set(6, 3, (Rr * A1) * (-0.25 * dt * dt));
    V.set(6, 6, Rr * (0.5 * dt));
    V.set(6, 9, (Rr * A1) * (-0.25 * dt * dt));
    V.set(9, 12, I3 * dt);
    V.set(12, 15, I3 * dt);
    s.J = F * s.J;
    Mat<15, 18> VN = V;
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 18; ++c) VN(r, c) *= nd[c];
    s.P = F * s.P * F.T() + VN * V.T();
    s.dp = dp1; s.dv = dv1; s.dq = dq1; s.dt += dt;
    a0 = a1; w0 = w1;
  }
  *out = s;
}

// whitening matrix of an IMU factor: W = chol(P)^-1 (lower). ||W r||^2 = r^T P^-1 r, i.e. the same cost,
// gradient and Gauss-Newton matrix as the reference's chol(P^-1)^T — the two differ by an
// orthogonal factor and the block carries no loss function (opt_be.cpp:416) — but formed without inverting P.
inline bool imu_whitening(const Mat<15, 15>& P, Mat<15, 15>* W) {
  Mat<15, 15> C = P;
  if (!chol_lower(C)) return false;
  *W = tri_lower_inverse(C);
  return true;
}

// R3 — robopt::imu::PreintegrationFactor (opt_be.cpp:415-416). Residual order
// [r_p, r_theta, r_v, r_ba, r_bg]; parameter order pose_i(6) sb_i(9) pose_j(6) sb_j(9) -> 15x30 Jacobian.
// The bias-correction Jacobian of r_theta uses the bias-corrected dq_c (exact derivative).
inline void imu_factor(const Preint& pi, const Mat<15, 15>& W, const Pose& Ti, const double* sbi, const Pose& Tj,
                       const double* sbj, double g, Mat<15, 1>* r, Mat<15, 30>* Jout) {
  const Vec3 vi = v3(sbi), bai = v3(sbi + 3), bgi = v3(sbi + 6);
  const Vec3 vj = v3(sbj), baj = v3(sbj + 3), bgj = v3(sbj + 6);
  const Vec3 dba = bai - pi.ba, dbg = bgi - pi.bg;
  const Mat3 Jp_ba = pi.J.block<3, 3>(0, 9), Jp_bg = pi.J.block<3, 3>(0, 12);
  const Mat3 Jq_bg = pi.J.block<3, 3>(3, 12);
  const Mat3 Jv_ba = pi.J.block<3, 3>(6, 9), Jv_bg = pi.J.block<3, 3>(6, 12);
  const Vec3 hq = Jq_bg * dbg * 0.5;
  const Quat dqc = (pi.dq * Quat(hq[0], hq[1], hq[2], 1.0)).normalized();
  const Vec3 dvc = pi.dv + Jv_ba * dba + Jv_bg * dbg;
  const Vec3 dpc = pi.dp + Jp_ba * dba + Jp_bg * dbg;
  const double dt = pi.dt;
  const Vec3 G = v3(0, 0, g);
  const Mat3 RiT = Ti.q.R().T();
  const Vec3 tp = RiT * (G * (0.5 * dt * dt) + Tj.p - Ti.p - vi * dt);
  const Vec3 tv = RiT * (G * dt + vj - vi);
  const Quat e = dqc.inv() * (Ti.q.inv() * Tj.q);
  Mat<15, 1> u;
  for (int k = 0; k < 3; ++k) {
    u[k] = tp[k] - dpc[k];
    u[3 + k] = 2.0 * e.vec()[k];
    u[6 + k] = tv[k] - dvc[k];
    u[9 + k] = baj[k] - bai[k];
    u[12 + k] = bgj[k] - bgi[k];
  }
  *r = W * u;
  if (!Jout) return;
  Mat<15, 30> A;
  const Mat3 I3 = Mat3::Identity();
  const Quat einv = e.inv();  // q_j^-1 q_i dq_c
  // pose_i
  A.set(0, 0, skew(tp));
  A.set(0, 3, -RiT);
  {
    // d r_theta / d dtheta_i = -(Lq(a) Rq(b))_vv for a = q_j^-1 q_i, b = dq_c: vector-vector 3x3 block of
    // the product of the full 4x4 (vector-first) left and right quaternion matrices
    const Quat a = Tj.q.inv() * Ti.q, b = dqc;
    Mat<4, 4> La, Rb;
    const Vec3 av = a.vec(), bv = b.vec();
    La.set(0, 0, I3 * a.w + skew(av));
    for (int k = 0; k < 3; ++k) { La(k, 3) = av[k]; La(3, k) = -av[k]; }
    La(3, 3) = a.w;
    Rb.set(0, 0, I3 * b.w - skew(bv));
    for (int k = 0; k < 3; ++k) { Rb(k, 3) = bv[k]; Rb(3, k) = -bv[k]; }
    Rb(3, 3) = b.w;
    const Mat<4, 4> LR = La * Rb;
    A.set(3, 0, -LR.block<3, 3>(0, 0));
  }
  A.set(6, 0, skew(tv));
  // sb_i : [v, ba, bg] at columns 6..14
  A.set(0, 6, RiT * (-dt));
  A.set(0, 9, -Jp_ba);
  A.set(0, 12, -Jp_bg);
  A.set(3, 12, -(quat_L3(einv) * Jq_bg));
  A.set(6, 6, -RiT);
  A.set(6, 9, -Jv_ba);
  A.set(6, 12, -Jv_bg);
  A.set(9, 9, -I3);
  A.set(12, 12, -I3);
  // pose_j at columns 15..20
  A.set(0, 18, RiT);
  A.set(3, 15, quat_L3(e));
  // sb_j at columns 21..29
  A.set(6, 21, RiT);
  A.set(9, 24, I3);
  A.set(12, 27, I3);
  *Jout = W * A;
}

}  // namespace covo
