// covo_solver.cpp — CPU oracle for the COVINS GBA / PGO hot path (problem level) + its C API.
//
// TEST INFRASTRUCTURE ONLY. May be imported / linked / executed solely by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg — as the checker, never as the thing shipped.
//
// PARITY UNPINNED (see covo_residuals.hpp header and DESIGN.md): Ceres 1.x / robopt_open / aslam_cv2 are
// not vendored in the reference and the reference holds no golden vectors for this path.
//
// Restates, on the flat IR of include/covgpu.h:
//   * what ceres::Solve(SPARSE_SCHUR, DOGLEG) does for the problems built at optimization_be.cpp:296-567
//     (GBA) and :836-1031 (PGO): linearise every residual block, eliminate the 3x3 landmark blocks by a
//     Schur complement, Cholesky-factor the reduced camera system, take a traditional-dogleg (or
//     Levenberg-Marquardt) trust-region step, accept / reject;
//   * the outlier evaluation of opt_be.cpp:270-290 and the PGO tail of :1046-1047, 1066-1081.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/covgpu.h"
#include "covo_residuals.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace covo {

struct State {
  std::vector<double> pose, sb, lm;
};

// Symmetric block pattern of the reduced camera system over keyframes (CSR, sorted columns, both (i,j) and (j,i)):
// a block (i,j) exists iff i == j, or i and j observe a common landmark, share an IMU factor or a between factor.
// The reference's SPARSE_SCHUR solver works on exactly this block structure (optimization_be.cpp:561).
struct BlockPattern {
  int K = 0;
  std::vector<int> ptr, col;
  int find(int i, int j) const {
    const int* b = col.data() + ptr[i];
    const int* e = col.data() + ptr[i + 1];
    const int* it = std::lower_bound(b, e, j);
    return (it != e && *it == j) ? (int)(it - col.data()) : -1;
  }
  size_t nnzb() const { return col.size(); }
};

struct Problem {
  const covgpu_problem* p;
  covgpu_options o;
  int D, K, L, O, I, E, n;
  bool vi;
  std::vector<Preint> pre;
  std::vector<Mat<15, 15>> preW;
  std::vector<int> obs_lm;
  std::vector<double> grav;  // gravity magnitude per IMU factor
  BlockPattern pat;
};

// Pluggable linear solver for the block-sparse reduced system (tests/ and tools/ install scipy's SuperLU through
// oracle/covo.py so that problems of BASELINE size are solved by a library independent of every Cholesky written in
// this repository). Returns 0 on success, non-zero if the matrix is not positive definite.
typedef int (*covo_sparse_solver_fn)(int n, int D, int K, const int* ptr, const int* col, const double* blocks, const double* rhs, double* x);
static covo_sparse_solver_fn g_sparse_solver = nullptr;
static int g_sparse_min_n = 0;

static void setup(Problem& P, const covgpu_problem* p, const covgpu_options* o, bool pgo) {
  P.p = p; P.o = *o;
  P.K = p->num_kf; P.L = pgo ? 0 : p->num_lm; P.O = pgo ? 0 : p->num_obs;
  P.vi = !pgo && !o->visual_only;
  P.I = P.vi ? p->num_imu : 0;
  P.E = p->num_edge;
  P.D = P.vi ? 15 : 6;
  P.n = P.D * P.K;
  P.obs_lm.assign(P.O, 0);
  for (int l = 0; l < P.L; ++l)
    for (int k = p->lm_obs_ptr[l]; k < p->lm_obs_ptr[l + 1]; ++k) P.obs_lm[k] = l;
  {  // block pattern
    std::vector<std::vector<int>> adj(P.K);
    for (int i = 0; i < P.K; ++i) adj[i].push_back(i);
    for (int l = 0; l < P.L; ++l)
      for (int a = p->lm_obs_ptr[l]; a < p->lm_obs_ptr[l + 1]; ++a)
        for (int b = p->lm_obs_ptr[l]; b < p->lm_obs_ptr[l + 1]; ++b)
          if (a != b) adj[p->obs_kf[a]].push_back(p->obs_kf[b]);
    for (int f = 0; f < P.I; ++f) { adj[p->imu_kf_i[f]].push_back(p->imu_kf_j[f]); adj[p->imu_kf_j[f]].push_back(p->imu_kf_i[f]); }
    for (int e = 0; e < P.E; ++e) { adj[p->edge_i[e]].push_back(p->edge_j[e]); adj[p->edge_j[e]].push_back(p->edge_i[e]); }
    P.pat.K = P.K; P.pat.ptr.assign(P.K + 1, 0); P.pat.col.clear();
    for (int i = 0; i < P.K; ++i) {
      std::sort(adj[i].begin(), adj[i].end());
      adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
      P.pat.col.insert(P.pat.col.end(), adj[i].begin(), adj[i].end());
      P.pat.ptr[i + 1] = (int)P.pat.col.size();
    }
  }
  // R2: repropagate every factor at the successor's current bias estimate (opt_be.cpp:387-396)
  P.pre.resize(P.I); P.preW.resize(P.I);
  P.grav.assign(P.I, o->gravity);
#pragma omp parallel for schedule(dynamic, 8)
  for (int f = 0; f < P.I; ++f) {
    // this factor's own calibration (keyframe_be.cpp:187-195), else the options' single set
    const double* q = p->imu_noise ? p->imu_noise + (size_t)5 * f : nullptr;
    const ImuNoise nz = q ? ImuNoise{q[0], q[1], q[2], q[3], q[4]} : ImuNoise{o->sigma_a, o->sigma_g, o->sigma_aw, o->sigma_gw, o->gravity};
    P.grav[f] = nz.g;
    const int j = p->imu_kf_j[f];
    const double* sbj = p->kf_speed_bias + 9 * j;
    const int s0 = p->imu_sample_ptr[f], s1 = p->imu_sample_ptr[f + 1];
    preintegrate(p->imu_first + 6 * f, p->imu_samples + 7 * s0, s1 - s0, v3(sbj + 3), v3(sbj + 6), nz, &P.pre[f]);
    if (!imu_whitening(P.pre[f].P, &P.preW[f])) P.preW[f] = Mat<15, 15>();
  }
}

static State initial_state(const Problem& P) {
  State s;
  s.pose.assign(P.p->kf_pose, P.p->kf_pose + 7 * P.K);
  if (P.vi) s.sb.assign(P.p->kf_speed_bias, P.p->kf_speed_bias + 9 * P.K);
  if (P.L) s.lm.assign(P.p->lm_pos, P.p->lm_pos + 3 * P.L);
  return s;
}

// ------------------------------------------------------------------ per-block evaluation
struct ObsLin { double r[2]; Mat<2, 6> Jp; Mat<2, 3> Jl; double cost; };

static void eval_obs(const Problem& P, const State& s, int k, bool jac, ObsLin* out) {
  const covgpu_problem* p = P.p;
  const int kf = p->obs_kf[k], l = P.obs_lm[k], cam = p->kf_cam[kf];
  const Pose Tws(&s.pose[7 * kf]), Tsc(p->cam_extr + 7 * cam);
  reproj(Tws, Tsc, v3(&s.lm[3 * l]), p->cam_intr + 4 * cam, p->cam_dist + 4 * cam, p->cam_dist_type[cam],
         p->obs_uv + 2 * k, p->obs_sigma[k], out->r, jac ? &out->Jp : nullptr, jac ? &out->Jl : nullptr);
  const double sq = cauchy(P.o.reproj_loss_a, out->r[0] * out->r[0] + out->r[1] * out->r[1], &out->cost);
  out->r[0] *= sq; out->r[1] *= sq;
  if (jac) {
    out->Jp = out->Jp * sq; out->Jl = out->Jl * sq;
    if (p->kf_fixed[kf]) out->Jp = Mat<2, 6>();
  }
}

static void eval_edge(const Problem& P, const State& s, int e, bool jac, Vec6* r, Mat<6, 6>* J1, Mat<6, 6>* J2, double* cost) {
  const covgpu_problem* p = P.p;
  const int i = p->edge_i[e], j = p->edge_j[e];
  Mat<6, 6> S;
  for (int k = 0; k < 36; ++k) S[k] = p->edge_sqrt_info[36 * e + k];
  between(Pose(&s.pose[7 * i]), Pose(&s.pose[7 * j]), Pose(p->edge_meas + 7 * e), S, r, jac ? J1 : nullptr, jac ? J2 : nullptr);
  const double sq = cauchy(p->edge_loss_a[e], r->squaredNorm(), cost);
  *r = *r * sq;
  if (jac) {
    *J1 = *J1 * sq; *J2 = *J2 * sq;
    if (p->kf_fixed[i]) *J1 = Mat<6, 6>();
    if (p->kf_fixed[j]) *J2 = Mat<6, 6>();
  }
}

static void eval_imu(const Problem& P, const State& s, int f, bool jac, Mat<15, 1>* r, Mat<15, 30>* J) {
  const covgpu_problem* p = P.p;
  const int i = p->imu_kf_i[f], j = p->imu_kf_j[f];
  imu_factor(P.pre[f], P.preW[f], Pose(&s.pose[7 * i]), &s.sb[9 * i], Pose(&s.pose[7 * j]), &s.sb[9 * j], P.grav[f], r,
             jac ? J : nullptr);
  if (jac) {
    if (p->kf_fixed[i]) for (int rr = 0; rr < 15; ++rr) for (int c = 0; c < 6; ++c) (*J)(rr, c) = 0;
    if (p->kf_fixed[j]) for (int rr = 0; rr < 15; ++rr) for (int c = 15; c < 21; ++c) (*J)(rr, c) = 0;
  }
}

// All sums below are formed in a fixed order (per-item values stored, then added serially), so the oracle's results
// do not depend on the OpenMP thread count of the box it runs on.
static double evaluate_cost(const Problem& P, const State& s) {
  std::vector<double> co(P.O), ci(P.I);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < P.O; ++k) { ObsLin ol; eval_obs(P, s, k, false, &ol); co[k] = ol.cost; }
#pragma omp parallel for schedule(static)
  for (int f = 0; f < P.I; ++f) { Mat<15, 1> r; eval_imu(P, s, f, false, &r, nullptr); ci[f] = 0.5 * r.squaredNorm(); }
  double cost = 0;
  for (double v : co) cost += v;
  for (double v : ci) cost += v;
  for (int e = 0; e < P.E; ++e) { Vec6 r; double c; eval_edge(P, s, e, false, &r, nullptr, nullptr, &c); cost += c; }
  return cost;
}

// ------------------------------------------------------------------ linearisation (A.6)
struct Lin {
  std::vector<double> H;    // nnzb x D x D  H_pp before Schur, block CSR on Problem::pat (full symmetric)
  std::vector<double> g;    // n      J_p^T r
  std::vector<double> Hll;  // L x 9
  std::vector<double> gl;   // L x 3
  std::vector<double> W;    // O x 18 (6x3)  J_p^T J_l
  std::vector<double> dp2;  // n      diag(H_pp)
  double cost = 0;
};

// H block (i,j), sub-block at (r0,c0) += B (R x C, leading dimension ldb)
static void add_block(const Problem& P, std::vector<double>& H, int i, int j, int r0, int c0, int R, int C, const double* B, int ldb) {
  const int D = P.D;
  double* blk = &H[(size_t)P.pat.find(i, j) * D * D];
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) blk[(r0 + r) * D + c0 + c] += B[r * ldb + c];
}

// dense n x n image of a block-CSR matrix (small problems, per-kernel test entry points)
static void expand_dense(const Problem& P, const std::vector<double>& Hb, std::vector<double>& M) {
  const int n = P.n, D = P.D;
  M.assign((size_t)n * n, 0.0);
  for (int i = 0; i < P.K; ++i)
    for (int q = P.pat.ptr[i]; q < P.pat.ptr[i + 1]; ++q) {
      const int j = P.pat.col[q];
      for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) M[(size_t)(D * i + r) * n + D * j + c] = Hb[(size_t)q * D * D + r * D + c];
    }
}

static void linearize(const Problem& P, const State& s, Lin& lin) {
  const int n = P.n, D = P.D;
  lin.H.assign(P.pat.nnzb() * D * D, 0.0);
  lin.g.assign(n, 0.0);
  lin.Hll.assign((size_t)P.L * 9, 0.0);
  lin.gl.assign((size_t)P.L * 3, 0.0);
  lin.W.assign((size_t)P.O * 18, 0.0);
  double cost = 0;
  std::vector<ObsLin> ol(P.O);
#pragma omp parallel for schedule(static)
  for (int l = 0; l < P.L; ++l) {
    Mat3 Hl; Vec3 gl;
    for (int k = P.p->lm_obs_ptr[l]; k < P.p->lm_obs_ptr[l + 1]; ++k) {
      eval_obs(P, s, k, true, &ol[k]);
      const Mat<3, 2> JlT = ol[k].Jl.T();
      Mat<2, 1> r; r[0] = ol[k].r[0]; r[1] = ol[k].r[1];
      Hl += JlT * ol[k].Jl;
      gl += JlT * r;
      const Mat<6, 3> W = ol[k].Jp.T() * ol[k].Jl;
      for (int q = 0; q < 18; ++q) lin.W[(size_t)k * 18 + q] = W[q];
    }
    for (int q = 0; q < 9; ++q) lin.Hll[(size_t)l * 9 + q] = Hl[q];
    for (int q = 0; q < 3; ++q) lin.gl[(size_t)l * 3 + q] = gl[q];
  }
  for (int k = 0; k < P.O; ++k) cost += ol[k].cost;  // fixed order
  // pose-side accumulation (serial: the oracle favours clarity)
  for (int k = 0; k < P.O; ++k) {
    const int kf = P.p->obs_kf[k];
    const Mat<6, 6> A = ol[k].Jp.T() * ol[k].Jp;
    Mat<2, 1> r; r[0] = ol[k].r[0]; r[1] = ol[k].r[1];
    const Mat<6, 1> b = ol[k].Jp.T() * r;
    add_block(P, lin.H, kf, kf, 0, 0, 6, 6, A.a, 6);
    for (int q = 0; q < 6; ++q) lin.g[D * kf + q] += b[q];
  }
  for (int f = 0; f < P.I; ++f) {
    Mat<15, 1> r; Mat<15, 30> J;
    eval_imu(P, s, f, true, &r, &J);
    cost += 0.5 * r.squaredNorm();
    const int i = P.p->imu_kf_i[f], j = P.p->imu_kf_j[f];
    const Mat<30, 30> A = J.T() * J;
    const Mat<30, 1> b = J.T() * r;
    add_block(P, lin.H, i, i, 0, 0, 15, 15, &A.a[0], 30);
    add_block(P, lin.H, i, j, 0, 0, 15, 15, &A.a[15], 30);
    add_block(P, lin.H, j, i, 0, 0, 15, 15, &A.a[15 * 30], 30);
    add_block(P, lin.H, j, j, 0, 0, 15, 15, &A.a[15 * 30 + 15], 30);
    for (int q = 0; q < 15; ++q) { lin.g[15 * i + q] += b[q]; lin.g[15 * j + q] += b[15 + q]; }
  }
  for (int e = 0; e < P.E; ++e) {
    Vec6 r; Mat<6, 6> J1, J2; double c;
    eval_edge(P, s, e, true, &r, &J1, &J2, &c);
    cost += c;
    const int i = P.p->edge_i[e], j = P.p->edge_j[e];
    const Mat<6, 6> A11 = J1.T() * J1, A12 = J1.T() * J2, A22 = J2.T() * J2;
    const Mat<6, 6> A21 = A12.T();
    const Mat<6, 1> b1 = J1.T() * r, b2 = J2.T() * r;
    add_block(P, lin.H, i, i, 0, 0, 6, 6, A11.a, 6);
    add_block(P, lin.H, i, j, 0, 0, 6, 6, A12.a, 6);
    add_block(P, lin.H, j, i, 0, 0, 6, 6, A21.a, 6);
    add_block(P, lin.H, j, j, 0, 0, 6, 6, A22.a, 6);
    for (int q = 0; q < 6; ++q) { lin.g[D * i + q] += b1[q]; lin.g[D * j + q] += b2[q]; }
  }
  lin.dp2.resize(n);
  for (int i = 0; i < P.K; ++i) {
    const double* blk = &lin.H[(size_t)P.pat.find(i, i) * D * D];
    for (int r = 0; r < D; ++r) lin.dp2[D * i + r] = blk[r * D + r];
  }
  lin.cost = cost;
}

static inline double clampd(double h) { return std::min(std::max(std::sqrt(std::max(h, 0.0)), 1e-6), 1e32); }

// damped Schur complement: S = H_pp + mu d_p^2 - sum_l W (H_ll + mu d_l^2)^-1 W^T ; b = -g_p + sum_l W (.)^-1 g_l
// S is block CSR on Problem::pat.
static void schur(const Problem& P, const Lin& lin, double mu, std::vector<double>& S, std::vector<double>& b,
                  std::vector<double>& HllInv) {
  const int n = P.n, D = P.D;
  S = lin.H;
  b.resize(n);
  for (int i = 0; i < P.K; ++i) {
    double* blk = &S[(size_t)P.pat.find(i, i) * D * D];
    for (int r = 0; r < D; ++r) {
      const int q = D * i + r;
      b[q] = -lin.g[q];
      const double d = clampd(lin.dp2[q]);
      if (lin.dp2[q] == 0.0) blk[r * D + r] = 1.0;  // constant / unconstrained dimension
      else blk[r * D + r] += mu * d * d;
    }
  }
  HllInv.assign((size_t)P.L * 9, 0.0);
  std::vector<double> Y((size_t)P.O * 18);
  // per-landmark inverse and Y = W (H_ll + mu d^2)^-1
#pragma omp parallel for schedule(static)
  for (int l = 0; l < P.L; ++l) {
    Mat3 Hl, Hi;
    for (int q = 0; q < 9; ++q) Hl[q] = lin.Hll[(size_t)l * 9 + q];
    for (int q = 0; q < 3; ++q) { const double d = clampd(Hl(q, q)); Hl(q, q) += mu * d * d; }
    if (!inv3_sym(Hl, Hi)) Hi = Mat3();
    for (int q = 0; q < 9; ++q) HllInv[(size_t)l * 9 + q] = Hi[q];
    for (int k = P.p->lm_obs_ptr[l]; k < P.p->lm_obs_ptr[l + 1]; ++k) {
      Mat<6, 3> W;
      for (int q = 0; q < 18; ++q) W[q] = lin.W[(size_t)k * 18 + q];
      const Mat<6, 3> Yk = W * Hi;
      for (int q = 0; q < 18; ++q) Y[(size_t)k * 18 + q] = Yk[q];
    }
  }
  // KF-major observation lists so that each thread owns whole block rows of S
  std::vector<int> kptr(P.K + 1, 0), kobs(P.O);
  for (int k = 0; k < P.O; ++k) kptr[P.p->obs_kf[k] + 1]++;
  for (int i = 0; i < P.K; ++i) kptr[i + 1] += kptr[i];
  {
    std::vector<int> cur(kptr.begin(), kptr.end() - 1);
    for (int k = 0; k < P.O; ++k) kobs[cur[P.p->obs_kf[k]]++] = k;
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < P.K; ++i) {
    for (int t = kptr[i]; t < kptr[i + 1]; ++t) {
      const int ka = kobs[t], l = P.obs_lm[ka];
      Mat<6, 3> Ya;
      for (int q = 0; q < 18; ++q) Ya[q] = Y[(size_t)ka * 18 + q];
      Vec3 gl = v3(&lin.gl[(size_t)l * 3]);
      const Mat<6, 1> yb = Ya * gl;
      for (int q = 0; q < 6; ++q) b[D * i + q] += yb[q];
      for (int kb = P.p->lm_obs_ptr[l]; kb < P.p->lm_obs_ptr[l + 1]; ++kb) {
        const int j = P.p->obs_kf[kb];
        Mat<6, 3> Wb;
        for (int q = 0; q < 18; ++q) Wb[q] = lin.W[(size_t)kb * 18 + q];
        const Mat<6, 6> YW = Ya * Wb.T();
        double* blk = &S[(size_t)P.pat.find(i, j) * D * D];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) blk[r * D + c] -= YW(r, c);
      }
    }
  }
}

// ------------------------------------------------------------------ dense Cholesky (lower, row-major)
static bool dense_cholesky(std::vector<double>& A, int n) {
  const int NB = 64;
  std::vector<double> PT;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    for (int j = k0; j < k0 + kb; ++j) {  // unblocked factor of the diagonal block
      double d = A[(size_t)j * n + j];
      for (int k = k0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0)) return false;
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k0 + kb; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    const int r0 = k0 + kb, m = n - r0;
    if (m <= 0) break;
#pragma omp parallel for schedule(static)
    for (int i = r0; i < n; ++i) {  // panel: A21 <- A21 L11^-T
      double* row = &A[(size_t)i * n + k0];
      for (int j = 0; j < kb; ++j) {
        double s = row[j];
        const double* lj = &A[(size_t)(k0 + j) * n + k0];
        for (int k = 0; k < j; ++k) s -= row[k] * lj[k];
        row[j] = s / lj[j];
      }
    }
    PT.assign((size_t)kb * m, 0.0);  // transposed panel so the trailing update streams contiguous rows
    for (int i = 0; i < m; ++i)
      for (int k = 0; k < kb; ++k) PT[(size_t)k * m + i] = A[(size_t)(r0 + i) * n + k0 + k];
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = r0; i < n; ++i) {  // trailing: A22 -= A21 A21^T (lower part)
      double* ci = &A[(size_t)i * n + r0];
      const double* ai = &A[(size_t)i * n + k0];
      const int len = i - r0 + 1;
      for (int k = 0; k < kb; ++k) {
        const double v = ai[k];
        const double* pt = &PT[(size_t)k * m];
        for (int j = 0; j < len; ++j) ci[j] -= v * pt[j];
      }
    }
  }
  return true;
}
static void dense_chol_solve(const std::vector<double>& Lf, int n, std::vector<double>& x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    const double* li = &Lf[(size_t)i * n];
    for (int k = 0; k < i; ++k) s -= li[k] * x[k];
    x[i] = s / li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    x[i] /= Lf[(size_t)i * n + i];
    const double v = x[i];
    const double* li = &Lf[(size_t)i * n];
    for (int k = 0; k < i; ++k) x[k] -= li[k] * v;
  }
}

// Solve S x = rhs for the block-CSR reduced system: the installed sparse solver (scipy SuperLU) for problems of at
// least g_sparse_min_n unknowns, else the dense Cholesky above on the expanded matrix. false: not positive definite.
static bool reduced_solve(const Problem& P, const std::vector<double>& S, std::vector<double>& bx) {
  if (g_sparse_solver != nullptr && P.n >= g_sparse_min_n) {
    std::vector<double> x(P.n);
    if (g_sparse_solver(P.n, P.D, P.K, P.pat.ptr.data(), P.pat.col.data(), S.data(), bx.data(), x.data()) != 0) return false;
    bx.swap(x);
    return true;
  }
  std::vector<double> M;
  expand_dense(P, S, M);
  if (!dense_cholesky(M, P.n)) return false;
  dense_chol_solve(M, P.n, bx);
  return true;
}

// full step (pose part dp given) -> landmark part: dl = (H_ll + mu d^2)^-1 (-g_l - sum W^T dp)
static void backsub(const Problem& P, const Lin& lin, const std::vector<double>& HllInv, const std::vector<double>& dp,
                    std::vector<double>& dl) {
  dl.assign((size_t)P.L * 3, 0.0);
#pragma omp parallel for schedule(static)
  for (int l = 0; l < P.L; ++l) {
    Vec3 t = -v3(&lin.gl[(size_t)l * 3]);
    for (int k = P.p->lm_obs_ptr[l]; k < P.p->lm_obs_ptr[l + 1]; ++k) {
      const int i = P.p->obs_kf[k];
      const double* W = &lin.W[(size_t)k * 18];
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 6; ++r) t[c] -= W[r * 3 + c] * dp[P.D * i + r];
    }
    Mat3 Hi;
    for (int q = 0; q < 9; ++q) Hi[q] = HllInv[(size_t)l * 9 + q];
    const Vec3 d = Hi * t;
    for (int q = 0; q < 3; ++q) dl[(size_t)l * 3 + q] = d[q];
  }
}

// v^T H v over the full (un-reduced, un-damped) Gauss-Newton matrix
static double quad_form(const Problem& P, const Lin& lin, const std::vector<double>& vp, const std::vector<double>& vl) {
  const int D = P.D;
  std::vector<double> qi(P.K, 0.0), ql(P.L, 0.0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P.K; ++i) {
    double acc = 0;
    for (int q = P.pat.ptr[i]; q < P.pat.ptr[i + 1]; ++q) {
      const int j = P.pat.col[q];
      const double* blk = &lin.H[(size_t)q * D * D];
      for (int r = 0; r < D; ++r) {
        if (vp[D * i + r] == 0.0) continue;
        double s = 0;
        for (int c = 0; c < D; ++c) s += blk[r * D + c] * vp[D * j + c];
        acc += vp[D * i + r] * s;
      }
    }
    qi[i] = acc;
  }
#pragma omp parallel for schedule(static)
  for (int l = 0; l < P.L; ++l) {
    double q2 = 0;
    const double* H = &lin.Hll[(size_t)l * 9];
    const double* v = &vl[(size_t)l * 3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) q2 += v[r] * H[r * 3 + c] * v[c];
    for (int k = P.p->lm_obs_ptr[l]; k < P.p->lm_obs_ptr[l + 1]; ++k) {
      const int i = P.p->obs_kf[k];
      const double* W = &lin.W[(size_t)k * 18];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c) q2 += 2.0 * vp[P.D * i + r] * W[r * 3 + c] * v[c];
    }
    ql[l] = q2;
  }
  double q = 0;
  for (double v : qi) q += v;
  for (double v : ql) q += v;
  return q;
}

static State apply_step(const Problem& P, const State& s, const std::vector<double>& dp, const std::vector<double>& dl) {
  State y = s;
  for (int i = 0; i < P.K; ++i) {
    pose_plus(Pose(&s.pose[7 * i]), &dp[P.D * i]).store(&y.pose[7 * i]);
    if (P.vi) for (int q = 0; q < 9; ++q) y.sb[9 * i + q] += dp[15 * i + 6 + q];
  }
  for (size_t q = 0; q < y.lm.size(); ++q) y.lm[q] += dl[q];
  return y;
}

static double x_norm(const Problem& P, const State& s) {
  double q = 0;
  for (int i = 0; i < P.K; ++i) {
    if (!P.p->kf_fixed[i]) for (int c = 0; c < 7; ++c) q += s.pose[7 * i + c] * s.pose[7 * i + c];
    if (P.vi) for (int c = 0; c < 9; ++c) q += s.sb[9 * i + c] * s.sb[9 * i + c];
  }
  for (double v : s.lm) q += v * v;
  return std::sqrt(q);
}

// ------------------------------------------------------------------ trust-region loop (A.6)
static int solve(Problem& P, State& x, covgpu_result* res) {
  const covgpu_options& o = P.o;
  const int n = P.n;
  Lin lin;
  linearize(P, x, lin);
  double cost = lin.cost;
  res->initial_cost = cost;
  double radius = o.initial_radius, mu = 1e-8, lm_df = 2.0;
  bool reuse = false;
  int term = 0, it = 0, accepted = 0;
  std::vector<double> S, b, HllInv, gn_p, gn_l, ghat_p, ghat_l, dpv(n), dlv((size_t)P.L * 3), stp, stl;
  double alpha = 0, dogleg_step_norm = 0;
  double t_lin = 0;
  auto dval_p = [&](int q) { return clampd(lin.dp2[q]); };
  auto dval_l = [&](int q) { return clampd(lin.Hll[(size_t)(q / 3) * 9 + (q % 3) * 4]); };
  for (; it < o.max_iterations; ++it) {
    double gmax = 0;
    for (int q = 0; q < n; ++q) gmax = std::max(gmax, std::fabs(lin.g[q]));
    for (double v : lin.gl) gmax = std::max(gmax, std::fabs(v));
    if (gmax <= o.gradient_tolerance) { term = 3; break; }
    bool ok = true;
    if (o.strategy == COVGPU_LM) {
      auto t0 = std::chrono::steady_clock::now();
      schur(P, lin, 1.0 / radius, S, b, HllInv);
      ok = reduced_solve(P, S, b);
      if (ok) {
        stp = b;
        backsub(P, lin, HllInv, stp, stl);
      }
      t_lin += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } else {
      if (!reuse) {
        // scaled gradient and Cauchy step length
        ghat_p.resize(n); ghat_l.resize((size_t)P.L * 3);
        std::vector<double> vp(n), vl((size_t)P.L * 3);
        double gg = 0;
        for (int q = 0; q < n; ++q) { const double d = dval_p(q); ghat_p[q] = lin.g[q] / d; vp[q] = ghat_p[q] / d; gg += ghat_p[q] * ghat_p[q]; }
        for (size_t q = 0; q < ghat_l.size(); ++q) { const double d = dval_l((int)q); ghat_l[q] = lin.gl[q] / d; vl[q] = ghat_l[q] / d; gg += ghat_l[q] * ghat_l[q]; }
        alpha = gg / quad_form(P, lin, vp, vl);
        auto t0 = std::chrono::steady_clock::now();
        ok = false;
        while (mu < 1.0) {
          schur(P, lin, mu, S, b, HllInv);
          if (reduced_solve(P, S, b)) { ok = true; break; }
          mu *= 10.0;
        }
        if (ok) {
          gn_p = b;
          backsub(P, lin, HllInv, gn_p, gn_l);
        }
        t_lin += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      if (ok) {
        double gn2 = 0, g2 = 0, gdot = 0;
        for (int q = 0; q < n; ++q) { const double d = dval_p(q), a = d * gn_p[q]; gn2 += a * a; g2 += ghat_p[q] * ghat_p[q]; gdot += ghat_p[q] * a; }
        for (size_t q = 0; q < gn_l.size(); ++q) { const double d = dval_l((int)q), a = d * gn_l[q]; gn2 += a * a; g2 += ghat_l[q] * ghat_l[q]; gdot += ghat_l[q] * a; }
        const double gn_norm = std::sqrt(gn2), g_norm = std::sqrt(g2);
        double cg, cn;  // step_hat = cg * ghat + cn * gn_hat
        if (gn_norm <= radius) { cg = 0; cn = 1; dogleg_step_norm = gn_norm; }
        else if (g_norm * alpha >= radius) { cg = -radius / g_norm; cn = 0; dogleg_step_norm = radius; }
        else {
          const double b_dot_a = -alpha * gdot, a_sq = alpha * alpha * g2;
          const double bma = gn2 - 2 * b_dot_a + a_sq, c = b_dot_a - a_sq;
          const double dd = std::sqrt(c * c + bma * (radius * radius - a_sq));
          const double beta = (c <= 0) ? (dd - c) / bma : (radius * radius - a_sq) / (dd + c);
          cg = -alpha * (1 - beta); cn = beta; dogleg_step_norm = radius;
        }
        stp.resize(n); stl.resize(gn_l.size());
        for (int q = 0; q < n; ++q) stp[q] = cg * ghat_p[q] / dval_p(q) + cn * gn_p[q];
        for (size_t q = 0; q < stl.size(); ++q) stl[q] = cg * ghat_l[q] / dval_l((int)q) + cn * gn_l[q];
      }
    }
    double model = 0;
    if (ok) {
      double gs = 0;
      for (int q = 0; q < n; ++q) gs += lin.g[q] * stp[q];
      for (size_t q = 0; q < stl.size(); ++q) gs += lin.gl[q] * stl[q];
      model = -(gs + 0.5 * quad_form(P, lin, stp, stl));
    }
    if (!ok || !(model > 0.0)) {  // invalid step
      if (o.strategy == COVGPU_LM) { radius /= lm_df; lm_df *= 2; } else { mu *= 10.0; reuse = false; }
      if (it < COVGPU_MAX_TRACE) { res->cost_trace[it] = cost; res->radius_trace[it] = radius; res->accepted_trace[it] = 0; }
      if (mu >= 1.0 && !ok) { term = 4; ++it; break; }
      continue;
    }
    double sn = 0;
    for (double v : stp) sn += v * v;
    for (double v : stl) sn += v * v;
    sn = std::sqrt(sn);
    if (sn <= o.parameter_tolerance * (x_norm(P, x) + o.parameter_tolerance)) { term = 2; break; }
    State xn = apply_step(P, x, stp, stl);
    const double cost_new = evaluate_cost(P, xn);
    const double rho = (cost - cost_new) / model;
    const bool acc = rho > o.min_relative_decrease;
    bool fn_conv = false;
    if (acc) {
      ++accepted;
      fn_conv = std::fabs(cost - cost_new) <= o.function_tolerance * cost;
      x = xn; cost = cost_new;
      if (o.strategy == COVGPU_LM) {
        const double t = 2 * rho - 1;
        radius = std::min(o.max_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
        lm_df = 2.0;
      } else {
        if (rho < 0.25) radius *= 0.5;
        if (rho > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
        mu = std::max(1e-8, 2.0 * mu / 10.0);
        reuse = false;
      }
      linearize(P, x, lin);
    } else {
      if (o.strategy == COVGPU_LM) { radius /= lm_df; lm_df *= 2; } else { radius *= 0.5; reuse = true; }
    }
    if (it < COVGPU_MAX_TRACE) { res->cost_trace[it] = cost; res->radius_trace[it] = radius; res->accepted_trace[it] = acc; }
    if (o.verbose) std::printf("[oracle] it %2d cost %.9e rho %.3f radius %.3e %s\n", it, cost, rho, radius, acc ? "ok" : "rej");
    if (fn_conv) { term = 1; ++it; break; }
  }
  res->iterations = it; res->accepted = accepted; res->termination = term;
  res->final_cost = cost;
  res->t_linear_solve_s = t_lin;
  return COVGPU_OK;
}

}  // namespace covo

// =================================================================== C API (ctypes-facing)
using namespace covo;

extern "C" {

int covo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void covo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static int run(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out, bool pgo) {
  Problem P;
  setup(P, p, opt, pgo);
  State x = initial_state(P);
  covgpu_result local;
  std::memset(&local, 0, sizeof(local));
  auto t0 = std::chrono::steady_clock::now();
  const int rc = solve(P, x, &local);
  local.t_solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::copy(x.pose.begin(), x.pose.end(), p->kf_pose);
  if (P.vi) std::copy(x.sb.begin(), x.sb.end(), p->kf_speed_bias);
  if (P.L) std::copy(x.lm.begin(), x.lm.end(), p->lm_pos);
  if (out) *out = local;
  return rc;
}
int covo_gba_solve(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out) { return run(opt, p, out, false); }
int covo_pgo_solve(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out) { return run(opt, p, out, true); }

int covo_reprojection_residual_norms(const covgpu_options* opt, const covgpu_problem* p, double* norms) {
  Problem P;
  covgpu_options o = *opt; o.visual_only = 1;
  setup(P, p, &o, false);
  State x = initial_state(P);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < P.O; ++k) { ObsLin ol; eval_obs(P, x, k, false, &ol); norms[k] = std::sqrt(ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1]); }
  return COVGPU_OK;
}

int covo_linearize_reprojection(const covgpu_options* opt, const covgpu_problem* p, double* r, double* Jp, double* Jl, double* cost) {
  Problem P;
  covgpu_options o = *opt; o.visual_only = 1;
  setup(P, p, &o, false);
  State x = initial_state(P);
  for (int k = 0; k < P.O; ++k) {
    ObsLin ol; eval_obs(P, x, k, true, &ol);
    r[2 * k] = ol.r[0]; r[2 * k + 1] = ol.r[1];
    for (int q = 0; q < 12; ++q) Jp[12 * k + q] = ol.Jp[q];
    for (int q = 0; q < 6; ++q) Jl[6 * k + q] = ol.Jl[q];
    cost[k] = ol.cost;
  }
  return COVGPU_OK;
}

int covo_preintegrate(const covgpu_options* opt, const covgpu_problem* p, double* delta, double* J, double* Pm) {
  Problem P;
  setup(P, p, opt, false);
  for (int f = 0; f < P.I; ++f) {
    const Preint& s = P.pre[f];
    double* d = delta + 11 * f;
    for (int k = 0; k < 3; ++k) { d[k] = s.dp[k]; d[7 + k] = s.dv[k]; }
    d[3] = s.dq.x; d[4] = s.dq.y; d[5] = s.dq.z; d[6] = s.dq.w; d[10] = s.dt;
    for (int q = 0; q < 225; ++q) { J[225 * f + q] = s.J[q]; Pm[225 * f + q] = s.P[q]; }
  }
  return COVGPU_OK;
}

int covo_linearize_imu(const covgpu_options* opt, const covgpu_problem* p, double* r, double* J) {
  Problem P;
  setup(P, p, opt, false);
  State x = initial_state(P);
  for (int f = 0; f < P.I; ++f) {
    Mat<15, 1> rr; Mat<15, 30> JJ;
    eval_imu(P, x, f, true, &rr, &JJ);
    for (int q = 0; q < 15; ++q) r[15 * f + q] = rr[q];
    for (int q = 0; q < 450; ++q) J[450 * f + q] = JJ[q];
  }
  return COVGPU_OK;
}

int covo_linearize_between(const covgpu_options* opt, const covgpu_problem* p, double* r, double* J, double* cost) {
  Problem P;
  setup(P, p, opt, true);
  State x = initial_state(P);
  for (int e = 0; e < P.E; ++e) {
    Vec6 rr; Mat<6, 6> J1, J2;
    eval_edge(P, x, e, true, &rr, &J1, &J2, &cost[e]);
    for (int q = 0; q < 6; ++q) r[6 * e + q] = rr[q];
    for (int a = 0; a < 6; ++a)
      for (int c = 0; c < 6; ++c) { J[72 * e + 12 * a + c] = J1(a, c); J[72 * e + 12 * a + 6 + c] = J2(a, c); }
  }
  return COVGPU_OK;
}

// Install (fn != NULL) or remove the sparse linear solver used for reduced systems with >= min_n unknowns.
void covo_set_sparse_solver(covo_sparse_solver_fn fn, int min_n) { g_sparse_solver = fn; g_sparse_min_n = min_n; }

// Block-sparse reduced system at the initial estimate: pattern sizes first (blocks == NULL), then the data.
// ptr[K+1], col[nnzb], blocks[nnzb][D*D], b[n]; returns nnzb.
int covo_schur_sparse(const covgpu_options* opt, const covgpu_problem* p, int pgo, double mu, int* ptr, int* col, double* blocks, double* b,
                      double* cost) {
  Problem P;
  setup(P, p, opt, pgo != 0);
  if (blocks == nullptr) return (int)P.pat.nnzb();
  State x = initial_state(P);
  Lin lin;
  linearize(P, x, lin);
  std::vector<double> Sv, bv, Hi;
  schur(P, lin, mu, Sv, bv, Hi);
  std::copy(P.pat.ptr.begin(), P.pat.ptr.end(), ptr);
  std::copy(P.pat.col.begin(), P.pat.col.end(), col);
  std::copy(Sv.begin(), Sv.end(), blocks);
  std::copy(bv.begin(), bv.end(), b);
  if (cost) *cost = lin.cost;
  return (int)P.pat.nnzb();
}

// H_ll (3x3 per landmark, undamped) at the estimate held in `p`: conditioning of each landmark for the parity criteria
int covo_landmark_hessians(const covgpu_options* opt, const covgpu_problem* p, double* Hll /* [L][9] */) {
  Problem P;
  covgpu_options o = *opt; o.visual_only = 1;
  setup(P, p, &o, false);
  State x = initial_state(P);
#pragma omp parallel for schedule(static)
  for (int l = 0; l < P.L; ++l) {
    Mat3 Hl;
    for (int k = p->lm_obs_ptr[l]; k < p->lm_obs_ptr[l + 1]; ++k) { ObsLin ol; eval_obs(P, x, k, true, &ol); Hl += ol.Jl.T() * ol.Jl; }
    for (int q = 0; q < 9; ++q) Hll[(size_t)l * 9 + q] = Hl[q];
  }
  return COVGPU_OK;
}

int covo_reduced_dim(const covgpu_options* opt, const covgpu_problem* p) { return (opt->visual_only ? 6 : 15) * p->num_kf; }

// pgo != 0: pose-graph problem (6 per KF, edges only)
int covo_schur(const covgpu_options* opt, const covgpu_problem* p, int pgo, double mu, double* S, double* b, double* cost) {
  Problem P;
  setup(P, p, opt, pgo != 0);
  State x = initial_state(P);
  Lin lin;
  linearize(P, x, lin);
  std::vector<double> Sv, bv, Hi;
  schur(P, lin, mu, Sv, bv, Hi);
  std::vector<double> M;
  expand_dense(P, Sv, M);
  std::copy(M.begin(), M.end(), S);
  std::copy(bv.begin(), bv.end(), b);
  *cost = lin.cost;
  return COVGPU_OK;
}

// dense reference: full (poses + landmarks) damped normal equations solved WITHOUT the Schur trick
// (golden check (5) of SURVEY.md §8c: dense-vs-Schur equality). Outputs dp[n], dl[3L].
int covo_dense_step(const covgpu_options* opt, const covgpu_problem* p, double mu, double* dp, double* dl) {
  Problem P;
  setup(P, p, opt, false);
  State x = initial_state(P);
  Lin lin;
  linearize(P, x, lin);
  const int n = P.n, N = n + 3 * P.L;
  std::vector<double> A((size_t)N * N, 0.0), rhs(N), Hd;
  expand_dense(P, lin.H, Hd);
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) A[(size_t)r * N + c] = Hd[(size_t)r * n + c];
    const double d = clampd(lin.dp2[r]);
    if (lin.dp2[r] == 0.0) A[(size_t)r * N + r] = 1.0; else A[(size_t)r * N + r] += mu * d * d;
    rhs[r] = -lin.g[r];
  }
  for (int l = 0; l < P.L; ++l) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) A[(size_t)(n + 3 * l + r) * N + n + 3 * l + c] = lin.Hll[(size_t)l * 9 + r * 3 + c];
      const double d = clampd(lin.Hll[(size_t)l * 9 + r * 4]);
      A[(size_t)(n + 3 * l + r) * N + n + 3 * l + r] += mu * d * d;
      rhs[n + 3 * l + r] = -lin.gl[(size_t)l * 3 + r];
    }
    for (int k = p->lm_obs_ptr[l]; k < p->lm_obs_ptr[l + 1]; ++k) {
      const int i = p->obs_kf[k];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c) {
          A[(size_t)(P.D * i + r) * N + n + 3 * l + c] += lin.W[(size_t)k * 18 + r * 3 + c];
          A[(size_t)(n + 3 * l + c) * N + P.D * i + r] += lin.W[(size_t)k * 18 + r * 3 + c];
        }
    }
  }
  if (!dense_cholesky(A, N)) return COVGPU_ERR_NUMERIC;
  dense_chol_solve(A, N, rhs);
  std::copy(rhs.begin(), rhs.begin() + n, dp);
  std::copy(rhs.begin() + n, rhs.end(), dl);
  return COVGPU_OK;
}

// Schur path of the same damped step (poses via reduced system, landmarks via back-substitution)
int covo_schur_step(const covgpu_options* opt, const covgpu_problem* p, double mu, double* dp, double* dl) {
  Problem P;
  setup(P, p, opt, false);
  State x = initial_state(P);
  Lin lin;
  linearize(P, x, lin);
  std::vector<double> S, b, Hi, dlv;
  schur(P, lin, mu, S, b, Hi);
  if (!reduced_solve(P, S, b)) return COVGPU_ERR_NUMERIC;
  backsub(P, lin, Hi, b, dlv);
  std::copy(b.begin(), b.end(), dp);
  std::copy(dlv.begin(), dlv.end(), dl);
  return COVGPU_OK;
}

int covo_solve_reduced(int n, const double* S, const double* b, double* x) {
  std::vector<double> A(S, S + (size_t)n * n), xv(b, b + n);
  if (!dense_cholesky(A, n)) return COVGPU_ERR_NUMERIC;
  dense_chol_solve(A, n, xv);
  std::copy(xv.begin(), xv.end(), x);
  return COVGPU_OK;
}

double covo_cost(const covgpu_options* opt, const covgpu_problem* p, int pgo) {
  Problem P;
  setup(P, p, opt, pgo != 0);
  State x = initial_state(P);
  return evaluate_cost(P, x);
}

// PGO tail (opt_be.cpp:1046-1047, 1066-1081)
int covo_pgo_reanchor(int num_kf, const double* pose_old, const double* pose_new, double* velocity, int num_lm,
                      const int* ref_kf, double* lm_pos) {
  if (velocity)
    for (int i = 0; i < num_kf; ++i) {
      const Mat3 Rn = Pose(pose_new + 7 * i).q.normalized().R(), Ro = Pose(pose_old + 7 * i).q.normalized().R();
      const Vec3 v = Rn * (Ro.T() * v3(velocity + 3 * i));
      for (int k = 0; k < 3; ++k) velocity[3 * i + k] = v[k];
    }
  for (int l = 0; l < num_lm; ++l) {
    const int k = ref_kf[l];
    if (k < 0) continue;
    const Pose To(pose_old + 7 * k), Tn(pose_new + 7 * k);
    const Vec3 ps = To.q.normalized().R().T() * (v3(lm_pos + 3 * l) - To.p);
    const Vec3 pw = Tn.q.normalized().R() * ps + Tn.p;
    for (int q = 0; q < 3; ++q) lm_pos[3 * l + q] = pw[q];
  }
  return COVGPU_OK;
}

// ---- raw residual functions for finite-difference tests of the oracle itself
void covo_pose_plus(const double* pose7, const double* d6, double* out7) { pose_plus(Pose(pose7), d6).store(out7); }
void covo_reproj_residual(const double* pose7, const double* extr7, const double* lm3, const double* intr4, const double* dist4,
                          int dist_type, const double* kp2, double sigma, double* r2, double* Jp12, double* Jl6) {
  Mat<2, 6> Jp; Mat<2, 3> Jl;
  reproj(Pose(pose7), Pose(extr7), v3(lm3), intr4, dist4, dist_type, kp2, sigma, r2, Jp12 ? &Jp : nullptr, Jl6 ? &Jl : nullptr);
  if (Jp12) for (int q = 0; q < 12; ++q) Jp12[q] = Jp[q];
  if (Jl6) for (int q = 0; q < 6; ++q) Jl6[q] = Jl[q];
}
void covo_between_residual(const double* pose1, const double* pose2, const double* meas7, const double* sqrt_info36, double* r6,
                           double* J1_36, double* J2_36) {
  Mat<6, 6> S, J1, J2; Vec6 r;
  for (int q = 0; q < 36; ++q) S[q] = sqrt_info36[q];
  between(Pose(pose1), Pose(pose2), Pose(meas7), S, &r, J1_36 ? &J1 : nullptr, J2_36 ? &J2 : nullptr);
  for (int q = 0; q < 6; ++q) r6[q] = r[q];
  if (J1_36) for (int q = 0; q < 36; ++q) J1_36[q] = J1[q];
  if (J2_36) for (int q = 0; q < 36; ++q) J2_36[q] = J2[q];
}
// one IMU factor: preintegrate samples at (ba_lin, bg_lin), evaluate at the given states.
// whiten = 0 returns the un-whitened residual / Jacobian (W = I).
void covo_imu_residual(const double* first6, const double* samples7, int n, const double* ba_lin, const double* bg_lin,
                       const double* noise5 /* sa sg saw sgw g */, const double* pose_i, const double* sb_i, const double* pose_j,
                       const double* sb_j, int whiten, double* r15, double* J450, double* delta11) {
  Preint pi;
  const ImuNoise nz{noise5[0], noise5[1], noise5[2], noise5[3], noise5[4]};
  preintegrate(first6, samples7, n, v3(ba_lin), v3(bg_lin), nz, &pi);
  Mat<15, 15> W = Mat<15, 15>::Identity();
  if (whiten) imu_whitening(pi.P, &W);
  Mat<15, 1> r; Mat<15, 30> J;
  imu_factor(pi, W, Pose(pose_i), sb_i, Pose(pose_j), sb_j, nz.g, &r, J450 ? &J : nullptr);
  for (int q = 0; q < 15; ++q) r15[q] = r[q];
  if (J450) for (int q = 0; q < 450; ++q) J450[q] = J[q];
  if (delta11) {
    for (int k = 0; k < 3; ++k) { delta11[k] = pi.dp[k]; delta11[7 + k] = pi.dv[k]; }
    delta11[3] = pi.dq.x; delta11[4] = pi.dq.y; delta11[5] = pi.dq.z; delta11[6] = pi.dq.w; delta11[10] = pi.dt;
  }
}

}  // extern "C"

extern "C" void covo_default_options(covgpu_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->strategy = COVGPU_DOGLEG;      // opt_be.cpp:564
  o->max_iterations = 10;           // config_backend.yaml:115 (opt.gba_iteration_limit)
  o->reproj_loss_a = 1.0;           // opt_be.cpp:302
  o->initial_radius = 1e4; o->max_radius = 1e16; o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
  // EuRoC.yaml:40-44 discretised at 200 Hz as in orb_slam3/src/Tracking.cc:1203-1211
  const double sf = std::sqrt(200.0);
  o->sigma_g = 1.7e-4 * sf; o->sigma_a = 2.0e-3 * sf; o->sigma_gw = 1.9393e-5 / sf; o->sigma_aw = 3.0e-3 / sf;
  o->gravity = 9.81;                // orb_slam3/include/ImuTypes.h:43
}
