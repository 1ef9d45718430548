// k_struct.hip — structure-exploiting solve of the damped reduced camera system.
//
// After landmark elimination the visual-inertial system couples, per keyframe k, a 6-dim pose block and a 9-dim
// speed-bias block. Landmarks and loop edges touch pose blocks only; a speed-bias block couples solely to its
// own pose and — through the two IMU factors of its keyframe — to the pose / speed-bias blocks of the chain
// neighbours (optimization_be.cpp:415-416: residual over pred pose, pred sb, kf pose, kf sb). Ordering the
// unknowns [all speed-bias blocks in IMU-chain order | all pose blocks] makes the speed-bias part A
// block-TRIDIAGONAL per agent, so it is eliminated exactly in O(K) small-block work; only the 6K x 6K pose
// system C' = C - B^T A^-1 B is factorised densely on the matrix cores: (6K)^3/3 instead of (15K)^3/3 flops,
// 15.6x fewer. This is what a fill-reducing sparse Cholesky (the reference's SPARSE_SCHUR -> CHOLMOD,
// optimization_be.cpp:561) obtains from the same sparsity; the result is the exact solve, just reordered.
//
//   sb_chain_factor   one workgroup per IMU chain, sequential over keyframes: L_kk L_kk^T = Ad_k - Lsub Lsub^T,
//                     Lsub_{k+1} = Ae_{k+1} L_kk^-T, and the rows of Y = L_A^-1 B and z = L_A^-1 b_s, stored
//                     transposed (Yt: pose rows x speed-bias columns) so that C -= Yt Yt^T is the ABT GEMM form
//   pose_rhs          b'_p = b_p - Yt z
//   (k_chol.hip)      C -= Yt Yt^T restricted to each tile pair's common chain segment; dense Cholesky of C'
//   sb_backsolve      x_s = A^-1 (b_s - B x_p) with the stored bidiagonal factor, one wave per chain
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

// gather the right-hand side: pose part into bp (chain-major, padded), speed-bias part into xs (chain order)
__global__ __launch_bounds__(256) void k_gather_rhs(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < P.npad) {
    double v = 0.0;
    if (q < 6 * P.K) { const int pos = q / 6, r = q - 6 * pos; v = P.bred[(size_t)P.D * P.pos_kf[pos] + r]; }
    P.bp[q] = v;
  }
  if (P.vi && q < 9 * P.K) { const int pos = q / 9, r = q - 9 * pos; P.xs[q] = P.bred[(size_t)15 * P.pos_kf[pos] + 6 + r]; }
}

__global__ __launch_bounds__(256) void k_scatter_solution(DevProblem P, double* __restrict__ dst) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.n) return;
  const int kf = q / P.D, r = q - kf * P.D, pos = P.perm[kf];
  dst[q] = (r < 6) ? P.bp[6 * pos + r] : P.xs[9 * pos + (r - 6)];
}

// One workgroup per chain. Sequential block-tridiagonal Cholesky + forward substitution of B's columns and b_s.
__global__ __launch_bounds__(256) void k_sb_chain_factor(DevProblem P) {
  __shared__ double sM[81], sL[81], sLinv[81], sSub[81], sNext[81];
  __shared__ double sz[9];
  const int tid = threadIdx.x;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  if (tid < 81) sSub[tid] = 0.0;
  if (tid < 9) sz[tid] = 0.0;
  __syncthreads();
  for (int pos = p0; pos < p1; ++pos) {
    const int t = pos - p0;
    // (1) M = Ad - Lsub Lsub^T
    if (tid < 81) {
      const int a = tid / 9, b = tid - 9 * a;
      double m = P.Ad[(size_t)81 * pos + tid];
      if (t > 0)
#pragma unroll
        for (int k = 0; k < 9; ++k) m -= sSub[9 * a + k] * sSub[9 * b + k];
      sM[tid] = m;
    }
    __syncthreads();
    // (2) 9x9 Cholesky and inverse of the factor (serial: 9x9 is latency-, not throughput-bound)
    if (tid == 0) {
      double L[81], X[81];
      bool ok = true;
      for (int k = 0; k < 81; ++k) { L[k] = 0.0; X[k] = 0.0; }
      for (int c = 0; c < 9; ++c) {
        double d = sM[10 * c];
        for (int k = 0; k < c; ++k) d -= L[9 * c + k] * L[9 * c + k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        d = sqrt(d);
        L[10 * c] = d;
        for (int r = c + 1; r < 9; ++r) {
          double s2 = sM[9 * r + c];
          for (int k = 0; k < c; ++k) s2 -= L[9 * r + k] * L[9 * c + k];
          L[9 * r + c] = s2 / d;
        }
      }
      for (int c = 0; c < 9; ++c) {
        X[10 * c] = 1.0 / L[10 * c];
        for (int r = c + 1; r < 9; ++r) {
          double s2 = 0.0;
          for (int k = c; k < r; ++k) s2 += L[9 * r + k] * X[9 * k + c];
          X[9 * r + c] = -s2 / L[10 * r];
        }
      }
      if (!ok) atomicOr(P.flag, 1);
      for (int k = 0; k < 81; ++k) { sL[k] = L[k]; sLinv[k] = X[k]; }
    }
    __syncthreads();
    if (tid < 81) {
      P.Ld[(size_t)81 * pos + tid] = sL[tid];
      P.Ldinv[(size_t)81 * pos + tid] = sLinv[tid];
      P.Lsub[(size_t)81 * pos + tid] = sSub[tid];
      // (4) next sub-diagonal block L_{pos+1,pos} = Ae_{pos+1} L_kk^-T   (computed now, published after step 3)
      double v = 0.0;
      if (pos + 1 < p1) {
        const int a = tid / 9, b = tid - 9 * a;
        const double* Ae = P.Ae + (size_t)81 * (pos + 1);
#pragma unroll
        for (int c = 0; c < 9; ++c) v += Ae[9 * a + c] * sLinv[9 * b + c];
      }
      sNext[tid] = v;
    }
    // (3) rows of Y and z at this position. Column (q, e) = pose dim e of chain position q; non-zero for q <= t+1.
    const int ncol = 6 * min(t + 2, p1 - p0);
    for (int jc = tid; jc <= ncol; jc += 256) {
      double v[9];
      if (jc == ncol) {  // right-hand side column
#pragma unroll
        for (int a = 0; a < 9; ++a) v[a] = P.xs[(size_t)9 * pos + a];
        if (t > 0)
#pragma unroll
          for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int k = 0; k < 9; ++k) v[a] -= sSub[9 * a + k] * sz[k];
      } else {
        const int q = jc / 6, e = jc - 6 * q;
        const double* Bblk = (q == t - 1) ? P.Bp : (q == t ? P.Bs : (q == t + 1 ? P.Bn : nullptr));
#pragma unroll
        for (int a = 0; a < 9; ++a) v[a] = Bblk ? Bblk[(size_t)54 * pos + 6 * a + e] : 0.0;
        if (t > 0 && q <= t) {
          const double* yprev = P.Yt + (size_t)(6 * (p0 + q) + e) * P.ldY + (size_t)9 * (pos - 1);
          double yp[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) yp[k] = yprev[k];
#pragma unroll
          for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int k = 0; k < 9; ++k) v[a] -= sSub[9 * a + k] * yp[k];
        }
      }
      double y[9];
#pragma unroll
      for (int a = 0; a < 9; ++a) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k <= a; ++k) s2 += sLinv[9 * a + k] * v[k];
        y[a] = s2;
      }
      if (jc == ncol) {
#pragma unroll
        for (int a = 0; a < 9; ++a) P.zs[(size_t)9 * pos + a] = y[a];
      } else {
        const int q = jc / 6, e = jc - 6 * q;
        double* yo = P.Yt + (size_t)(6 * (p0 + q) + e) * P.ldY + (size_t)9 * pos;
#pragma unroll
        for (int a = 0; a < 9; ++a) yo[a] = y[a];
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid < 81) sSub[tid] = sNext[tid];
    if (tid < 9) sz[tid] = P.zs[(size_t)9 * pos + tid];
    __syncthreads();
  }
}

// b'_p[row] = b_p[row] - sum_k Yt[row][k] z[k] over the row's chain segment. One wave per row.
__global__ __launch_bounds__(256) void k_pose_rhs(DevProblem P) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 6 * P.K) return;
  const int pos = row / 6;
  const int kb = 9 * max(pos - 1, 0), ke = 9 * P.pos_chain_end[pos];
  const double* y = P.Yt + (size_t)row * P.ldY;
  double acc = 0.0;
  for (int k = kb + lane; k < ke; k += 64) acc += y[k] * P.zs[k];
  acc = wave_sum(acc);
  if (lane == 0) P.bp[row] -= acc;
}

// x_s = A^-1 (b_s - B x_p): w = b_s - B x_p, forward with (Ldinv, Lsub), backward with their transposes.
// One 64-lane workgroup per chain; lanes 0..8 each own one row of the current 9-vector.
__global__ __launch_bounds__(64) void k_sb_backsolve(DevProblem P) {
  __shared__ double su[9], sprev[9];
  const int lane = threadIdx.x;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  if (lane < 9) sprev[lane] = 0.0;
  __syncthreads();
  for (int pos = p0; pos < p1; ++pos) {  // forward: u_pos = Linv (w_pos - Lsub u_{pos-1}), stored in xs
    double w = 0.0;
    if (lane < 9) {
      w = P.xs[(size_t)9 * pos + lane];
      const double* xp_self = P.bp + 6 * pos;
#pragma unroll
      for (int e = 0; e < 6; ++e) w -= P.Bs[(size_t)54 * pos + 6 * lane + e] * xp_self[e];
      if (pos > p0)
#pragma unroll
        for (int e = 0; e < 6; ++e) w -= P.Bp[(size_t)54 * pos + 6 * lane + e] * P.bp[6 * (pos - 1) + e];
      if (pos + 1 < p1)
#pragma unroll
        for (int e = 0; e < 6; ++e) w -= P.Bn[(size_t)54 * pos + 6 * lane + e] * P.bp[6 * (pos + 1) + e];
      if (pos > p0)
#pragma unroll
        for (int k = 0; k < 9; ++k) w -= P.Lsub[(size_t)81 * pos + 9 * lane + k] * sprev[k];
      su[lane] = w;
    }
    __syncthreads();
    if (lane < 9) {
      double u = 0.0;
      for (int k = 0; k <= lane; ++k) u += P.Ldinv[(size_t)81 * pos + 9 * lane + k] * su[k];
      P.xs[(size_t)9 * pos + lane] = u;
      sprev[lane] = u;
    }
    __syncthreads();
  }
  if (lane < 9) sprev[lane] = 0.0;
  __syncthreads();
  for (int pos = p1 - 1; pos >= p0; --pos) {  // backward: x_pos = Linv^T (u_pos - Lsub_{pos+1}^T x_{pos+1})
    if (lane < 9) {
      double w = P.xs[(size_t)9 * pos + lane];
      if (pos + 1 < p1)
#pragma unroll
        for (int k = 0; k < 9; ++k) w -= P.Lsub[(size_t)81 * (pos + 1) + 9 * k + lane] * sprev[k];
      su[lane] = w;
    }
    __syncthreads();
    if (lane < 9) {
      double x = 0.0;
      for (int k = lane; k < 9; ++k) x += P.Ldinv[(size_t)81 * pos + 9 * k + lane] * su[k];
      P.xs[(size_t)9 * pos + lane] = x;
      sprev[lane] = x;
    }
    __syncthreads();
  }
}

void launch_structured_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax) {
  const int cnt = P.npad > 9 * P.K ? P.npad : 9 * P.K;
  hipLaunchKernelGGL(k_gather_rhs, dim3((cnt + 255) / 256), dim3(256), 0, st, P);
  if (P.vi) {
    hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_pose_rhs, dim3((6 * P.K + 3) / 4), dim3(256), 0, st, P);
    launch_yty_update(P, st);
  }
  dense_cholesky_solve_raw(P.Sred, P.bp, P.Linv, P.flag, P.npad, st, ax);
  if (P.vi) hipLaunchKernelGGL(k_sb_backsolve, dim3(P.nchains), dim3(64), 0, st, P);
  hipLaunchKernelGGL(k_scatter_solution, dim3((P.n + 255) / 256), dim3(256), 0, st, P, dst);
}

}  // namespace covgpu
