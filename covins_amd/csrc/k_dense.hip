// k_dense.hip — dense FP64 reduced-camera-system solve on the gfx950 matrix cores, plus the flat vector
// kernels of the trust-region step.
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
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;        // K chunk staged through LDS
constexpr int LDT = KC + 1;   // LDS pitch (doubles): odd pitch -> conflict-free fragment reads

enum { MODE_SYRK = 0, MODE_TRSM = 1 };

// C[i][j] (op)= sum_k A[i][k] B[j][k] for one 128x128 tile, K = 128.
//   SYRK: tile (ti, tj) of the trailing matrix starting at row/col r0; A/B = panel rows r0+ti*128.. / r0+tj*128.., cols k0..
//   TRSM: row tile ti of the panel below the diagonal block; B = Linv (128x128, pitch 128); result overwrites A.
template <int MODE>
__global__ __launch_bounds__(256) void k_gemm_abt(double* __restrict__ M, size_t ld, int k0, int r0, const double* __restrict__ Linv) {
  int ti, tj;
  if (MODE == MODE_SYRK) {
    ti = blockIdx.y; tj = blockIdx.x;
    if (tj > ti) return;
  } else {
    ti = blockIdx.x; tj = 0;
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];  // 2 x [128][33] doubles = 66 KiB (> 64 KiB static limit)
  double (*sA)[LDT] = reinterpret_cast<double (*)[LDT]>(smem);
  double (*sB)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + kTile * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const double* Ag = M + (size_t)(r0 + ti * kTile) * ld + k0;
  const double* Bg;
  size_t ldb;
  if (MODE == MODE_SYRK) { Bg = M + (size_t)(r0 + tj * kTile) * ld + k0; ldb = ld; }
  else { Bg = Linv; ldb = kTile; }
  // staging map: 16 lanes cover one 32-double row segment (256 B contiguous), 16 rows per pass, 8 passes
  const int c2 = (tid & 15) * 2, rbase = tid >> 4;
  double2 pa[8], pb[8];
  auto gload = [&](int kc) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = rbase + 16 * it;
      pa[it] = *reinterpret_cast<const double2*>(Ag + (size_t)row * ld + kc + c2);
      pb[it] = *reinterpret_cast<const double2*>(Bg + (size_t)row * ldb + kc + c2);
    }
  };
  v4f64 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
  gload(0);
  const int fr = lane & 15, fk = lane >> 4;
  for (int kc = 0; kc < kTile; kc += KC) {
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = rbase + 16 * it;
      sA[row][c2] = pa[it].x; sA[row][c2 + 1] = pa[it].y;
      sB[row][c2] = pb[it].x; sB[row][c2 + 1] = pb[it].y;
    }
    __syncthreads();
    if (kc + KC < kTile) gload(kc + KC);  // prefetch the next chunk while the matrix cores work
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = sA[wr * 64 + t * 16 + fr][kk + fk];
        b[t] = sB[wc * 64 + t * 16 + fr][kk + fk];
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
  }
  // epilogue. f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* Cg = (MODE == MODE_SYRK) ? M + (size_t)(r0 + ti * kTile) * ld + (r0 + tj * kTile) : M + (size_t)(r0 + ti * kTile) * ld + k0;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int row = wr * 64 + tm * 16 + fk + 4 * rg, col = wc * 64 + tn * 16 + fr;
        double* p = Cg + (size_t)row * ld + col;
        if (MODE == MODE_SYRK) *p -= acc[tm][tn][rg]; else *p = acc[tm][tn][rg];
      }
}

// Factor the 128x128 diagonal block at (k0,k0) in LDS (lower Cholesky) and form its inverse.
// L (lower incl. diagonal) is written back into M; L^-1 (lower, zeros above) goes to Linv_out [128][128].
__global__ __launch_bounds__(256) void k_potrf_inv(double* __restrict__ M, size_t ld, int k0, double* __restrict__ Linv_out, int* flag) {
  extern __shared__ __attribute__((aligned(16))) double s[];  // [128][129]
  constexpr int PT = kTile + 1;
  const int tid = threadIdx.x;
  double* Mg = M + (size_t)k0 * ld + k0;
  for (int idx = tid; idx < kTile * kTile; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    s[r * PT + c] = (c <= r) ? Mg[(size_t)r * ld + c] : 0.0;
  }
  const int ty = tid >> 4, tx = tid & 15;
  for (int j = 0; j < kTile; ++j) {
    __syncthreads();
    double d = s[j * PT + j];
    if (!(d > 0.0)) { if (tid == 0) atomicOr(flag, 1); d = 1.0; }
    const double sd = sqrt(d), inv = 1.0 / sd;
    __syncthreads();
    if (tid < kTile) {
      if (tid == j) s[j * PT + j] = sd;
      else if (tid > j) s[tid * PT + j] *= inv;
    }
    __syncthreads();
    for (int r = j + 1 + ty; r < kTile; r += 16) {
      const double lr = s[r * PT + j];
      for (int c = j + 1 + tx; c <= r; c += 16) s[r * PT + c] -= lr * s[c * PT + j];
    }
  }
  __syncthreads();
  // inverse, one thread per column c; X[r][c] (r > c) is kept in the unused upper triangle at s[c][r]
  __shared__ double xd[kTile];
  if (tid < kTile) {
    const int c = tid;
    const double xcc = 1.0 / s[c * PT + c];
    xd[c] = xcc;
    for (int r = c + 1; r < kTile; ++r) {
      double sum = s[r * PT + c] * xcc;
      for (int k = c + 1; k < r; ++k) sum += s[r * PT + k] * s[c * PT + k];
      s[c * PT + r] = -sum / s[r * PT + r];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < kTile * kTile; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    if (c <= r) Mg[(size_t)r * ld + c] = s[r * PT + c];
    Linv_out[idx] = (c < r) ? s[c * PT + r] : (c == r ? xd[c] : 0.0);
  }
}

// forward substitution step for panel p:  y_p = Linv_p b_p ; b[rows below] -= L[rows, panel p] y_p
__global__ __launch_bounds__(256) void k_fwd_step(const double* __restrict__ M, size_t ld, int p, const double* __restrict__ Linv,
                                                   double* __restrict__ b, double* __restrict__ y) {
  __shared__ double sy[kTile];
  __shared__ double sb_[kTile];
  const int tid = threadIdx.x, k0 = p * kTile;
  if (tid < kTile) sb_[tid] = b[k0 + tid];
  __syncthreads();
  if (tid < kTile) {
    const double* Lr = Linv + (size_t)tid * kTile;
    double s2 = 0.0;
    for (int k = 0; k <= tid; ++k) s2 += Lr[k] * sb_[k];
    sy[tid] = s2;
    if (blockIdx.x == 0) y[k0 + tid] = s2;
  }
  __syncthreads();
  // each block updates 128 rows below the panel; 2 threads per row, each half of the 128 columns
  const int row = k0 + kTile + blockIdx.x * kTile + (tid >> 1), half = tid & 1;
  if (row >= (int)ld) return;  // last panel: nothing below (no barrier follows)
  const double* Lr = M + (size_t)row * ld + k0 + half * 64;
  double s2 = 0.0;
#pragma unroll 8
  for (int k = 0; k < 64; ++k) s2 += Lr[k] * sy[half * 64 + k];
  s2 += __shfl_xor(s2, 1, 64);
  if (half == 0) b[row] -= s2;
}

// backward substitution step for panel p:  x_p = Linv_p^T y_p ; y[cols left of the panel] -= L[panel rows, cols]^T x_p
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ M, size_t ld, int p, const double* __restrict__ Linv,
                                                   double* __restrict__ y, double* __restrict__ x) {
  __shared__ double sx[kTile];
  __shared__ double sy[kTile];
  const int tid = threadIdx.x, k0 = p * kTile;
  if (tid < kTile) sy[tid] = y[k0 + tid];
  __syncthreads();
  if (tid < kTile) {
    double s2 = 0.0;
    for (int j = tid; j < kTile; ++j) s2 += Linv[(size_t)j * kTile + tid] * sy[j];
    sx[tid] = s2;
    if (blockIdx.x == 0) x[k0 + tid] = s2;
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col < k0) {
    const double* Lc = M + (size_t)k0 * ld + col;
    double s2 = 0.0;
#pragma unroll 8
    for (int r = 0; r < kTile; ++r) s2 += Lc[(size_t)r * ld] * sx[r];
    y[col] -= s2;
  }
}

void dense_cholesky_solve_raw(double* S, double* b, double* Linv, int* flag, int npad, hipStream_t st, hipEvent_t* syrk_events) {
  const int T = npad / kTile;
  const size_t ld = (size_t)npad;
  const size_t lds_potrf = (size_t)kTile * (kTile + 1) * sizeof(double);
  const size_t lds_gemm = (size_t)2 * kTile * LDT * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_potrf);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_TRSM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
    attr_set = true;
  }
  for (int p = 0; p < T; ++p) {
    const int k0 = p * kTile, rem = T - p - 1;
    double* Li = Linv + (size_t)p * kTile * kTile;
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), lds_potrf, st, S, ld, k0, Li, flag);
    if (rem > 0) {
      hipLaunchKernelGGL(k_gemm_abt<MODE_TRSM>, dim3(rem), dim3(256), lds_gemm, st, S, ld, k0, k0 + kTile, Li);
      if (syrk_events) (void)hipEventRecord(syrk_events[2 * p], st);
      hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK>, dim3(rem, rem), dim3(256), lds_gemm, st, S, ld, k0, k0 + kTile, nullptr);
      if (syrk_events) (void)hipEventRecord(syrk_events[2 * p + 1], st);
    }
  }
  // L y = b (y kept in b's panel slots via a scratch alias: y and x live in `b` itself — each panel slot is
  // final before it is overwritten, see k_fwd_step / k_bwd_step)
  for (int p = 0; p < T; ++p) {
    const int rem = T - p - 1;
    hipLaunchKernelGGL(k_fwd_step, dim3(rem > 0 ? rem : 1), dim3(256), 0, st, S, ld, p, Linv + (size_t)p * kTile * kTile, b, b + npad);
  }
  for (int p = T - 1; p >= 0; --p) {
    const int nb = (p * kTile + 255) / 256;
    hipLaunchKernelGGL(k_bwd_step, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, S, ld, p, Linv + (size_t)p * kTile * kTile, b + npad, b);
  }
}

void launch_dense_cholesky_solve(const DevProblem& P, hipStream_t st, hipEvent_t* syrk_events) {
  dense_cholesky_solve_raw(P.Sred, P.bred, P.Linv, P.flag, P.npad, st, syrk_events);
}

// ------------------------------------------------------------------------------------------- vector kernels
__global__ __launch_bounds__(256) void k_finalize_diag(DevProblem P, double mu) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.npad) return;
  double* d = P.Sred + (size_t)q * P.npad + q;
  if (q >= P.n) { *d = 1.0; P.bred[q] = 0.0; return; }
  const double h = P.hdiag[q];
  if (h == 0.0) { *d = 1.0; P.bred[q] = 0.0; }  // constant / unconstrained dimension (A.6)
  else { const double c = clamp_diag(h); *d += mu * c * c; }
}

COV_DEV void block_reduce_atomic(double v, double* dst, bool is_max = false) {
  if (is_max) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) {
      // non-negative doubles order like their bit patterns
      atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v));
    }
  } else {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(dst, v);
  }
}

// GG = |g/d|^2, GN2 = |d gn|^2, GDOT = g.gn, GMAX = max|g|
__global__ __launch_bounds__(256) void k_dogleg_stats(DevProblem P) {
  double gg = 0, gn2 = 0, gd = 0, gm = 0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double g = P.grad[q], d = clamp_diag(P.hdiag[q]), s = P.gn[q];
    gg += (g / d) * (g / d); gn2 += (d * s) * (d * s); gd += g * s; gm = fmax(gm, fabs(g));
  }
  block_reduce_atomic(gg, &P.scal[SC_GG]);
  block_reduce_atomic(gn2, &P.scal[SC_GN2]);
  block_reduce_atomic(gd, &P.scal[SC_GDOT]);
  block_reduce_atomic(gm, &P.scal[SC_GMAX], true);
}

__global__ __launch_bounds__(256) void k_cauchy_vec(DevProblem P) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double d = clamp_diag(P.hdiag[q]);
    P.vtmp[q] = P.grad[q] / (d * d);
  }
}

// step = cg * g / d^2 + cn * gn ;  GS = g.step, SN2 = |step|^2
__global__ __launch_bounds__(256) void k_combine_step(DevProblem P, double cg, double cn) {
  double gs = 0, sn = 0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P.N; q += gridDim.x * blockDim.x) {
    const double g = P.grad[q], d = clamp_diag(P.hdiag[q]);
    double s = cn * P.gn[q];
    if (cg != 0.0) s += cg * g / (d * d);
    P.step[q] = s;
    gs += g * s; sn += s * s;
  }
  block_reduce_atomic(gs, &P.scal[SC_GS]);
  block_reduce_atomic(sn, &P.scal[SC_SN2]);
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
    if (!P.fixed[t]) for (int k = 0; k < 7; ++k) acc += P.pose[7 * t + k] * P.pose[7 * t + k];
    if (P.vi) for (int k = 0; k < 9; ++k) acc += P.sb[9 * t + k] * P.sb[9 * t + k];
  }
  for (int q = t0; q < 3 * P.L; q += stride) acc += P.lm[q] * P.lm[q];
  block_reduce_atomic(acc, &P.scal[SC_XN2]);
}

static inline int vec_grid(int n) {
  const int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

void launch_finalize_diag(const DevProblem& P, double mu, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize_diag, dim3((P.npad + 255) / 256), dim3(256), 0, st, P, mu);
}
void launch_zero_system(const DevProblem& P, hipStream_t st) {
  hipMemsetAsync(P.Sred, 0, (size_t)P.npad * P.npad * sizeof(double), st);
  hipMemsetAsync(P.bred, 0, (size_t)2 * P.npad * sizeof(double), st);
  hipMemsetAsync(P.grad, 0, (size_t)P.N * sizeof(double), st);
  hipMemsetAsync(P.hdiag, 0, (size_t)P.N * sizeof(double), st);
  hipMemsetAsync(P.scal + SC_COST, 0, sizeof(double), st);
  hipMemsetAsync(P.flag, 0, sizeof(int), st);
}
void launch_dogleg_stats(const DevProblem& P, hipStream_t st) {
  hipMemsetAsync(P.scal + SC_GG, 0, 4 * sizeof(double), st);
  hipLaunchKernelGGL(k_dogleg_stats, dim3(vec_grid(P.N)), dim3(256), 0, st, P);
}
void launch_cauchy_vec(const DevProblem& P, hipStream_t st) {
  hipLaunchKernelGGL(k_cauchy_vec, dim3(vec_grid(P.N)), dim3(256), 0, st, P);
}
void launch_combine_step(const DevProblem& P, double cg, double cn, hipStream_t st) {
  hipMemsetAsync(P.scal + SC_GS, 0, 2 * sizeof(double), st);
  hipLaunchKernelGGL(k_combine_step, dim3(vec_grid(P.N)), dim3(256), 0, st, P, cg, cn);
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
  hipMemsetAsync(P.scal + SC_XN2, 0, sizeof(double), st);
  const int n = P.K > 3 * P.L ? P.K : 3 * P.L;
  hipLaunchKernelGGL(k_xnorm, dim3(vec_grid(n)), dim3(256), 0, st, P);
}

}  // namespace covgpu
