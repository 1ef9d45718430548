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
//   sb_chain_factor   one wave per IMU chain, sequential over keyframes: L_kk L_kk^T = Ad_k - Lsub Lsub^T,
//                     Lsub_{k+1} = Ae_{k+1} L_kk^-T, z = L_A^-1 b_s
//   sb_chain_cols     Y = L_A^-1 B, one thread per pose dimension marching down its chain (K-major, coalesced)
//   pose_rhs          b'_p = b_p - Y^T z
//   (k_chol.hip)      C -= Y^T Y restricted to each tile pair's common chain segment; dense Cholesky of C'
//   sb_backsolve      x_s = A^-1 (b_s - B x_p) with the stored bidiagonal factor, one wave per chain
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

// gather the right-hand side: pose part into bp (chain-major, padded), speed-bias part into xs (chain order)
// which: 0 = pose part, 1 = speed-bias part, 2 = both
__global__ __launch_bounds__(256) void k_gather_rhs(DevProblem P, int which) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (which != 1 && q < P.npad) {
    double v = 0.0;
    if (q < 6 * P.K) { const int pos = q / 6, r = q - 6 * pos; v = P.bred[(size_t)P.D * P.pos_kf[pos] + r]; }
    P.bp[q] = v;
  }
  if (which != 0 && P.vi && q < 9 * P.K) { const int pos = q / 9, r = q - 9 * pos; P.xs[q] = P.bred[(size_t)15 * P.pos_kf[pos] + 6 + r]; }
}

__global__ __launch_bounds__(256) void k_scatter_solution(DevProblem P, double* __restrict__ dst) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.n) return;
  const int kf = q / P.D, r = q - kf * P.D, pos = P.perm[kf];
  dst[q] = (r < 6) ? P.bp[6 * pos + r] : P.xs[9 * pos + (r - 6)];
}

COV_DEV double rdlane64(double v, int srclane) {  // broadcast from a wave-uniform lane through SGPRs
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), srclane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One 64-lane workgroup (a single wave) per chain: the sequential part only — block-bidiagonal Cholesky of the
// speed-bias system and z = L_A^-1 b_s. Per keyframe: 9x9 Cholesky in registers (row per lane, v_readlane
// broadcasts), inverse by columns from LDS, next blocks prefetched from HBM one step ahead.
__global__ __launch_bounds__(64) void k_sb_chain_factor(DevProblem P) {
  __shared__ double sM[81], sX[81], sSub[81];
  __shared__ double sz[9], sv[9];
  const int lane = threadIdx.x;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const int e0 = lane, e1 = lane + 64;  // the two matrix entries this lane owns (e1 valid for lanes < 17)
  const bool has1 = e1 < 81;
  for (int e = lane; e < 81; e += 64) sSub[e] = 0.0;
  if (lane < 9) sz[lane] = 0.0;
  // prefetch registers: Ad of the current position, Ae of the next, b_s of the current
  double ad0 = P.Ad[(size_t)81 * p0 + e0], ad1 = has1 ? P.Ad[(size_t)81 * p0 + e1] : 0.0;
  double ae0 = 0.0, ae1 = 0.0;
  if (p0 + 1 < p1) { ae0 = P.Ae[(size_t)81 * (p0 + 1) + e0]; ae1 = has1 ? P.Ae[(size_t)81 * (p0 + 1) + e1] : 0.0; }
  double bs = (lane < 9) ? P.xs[(size_t)9 * p0 + lane] : 0.0;
  __syncthreads();
  for (int pos = p0; pos < p1; ++pos) {
    // issue next step's loads first
    double nad0 = 0.0, nad1 = 0.0, nae0 = 0.0, nae1 = 0.0, nbs = 0.0;
    if (pos + 1 < p1) {
      nad0 = P.Ad[(size_t)81 * (pos + 1) + e0]; nad1 = has1 ? P.Ad[(size_t)81 * (pos + 1) + e1] : 0.0;
      nbs = (lane < 9) ? P.xs[(size_t)9 * (pos + 1) + lane] : 0.0;
    }
    if (pos + 2 < p1) { nae0 = P.Ae[(size_t)81 * (pos + 2) + e0]; nae1 = has1 ? P.Ae[(size_t)81 * (pos + 2) + e1] : 0.0; }
    // M = Ad - Lsub Lsub^T
    {
      const int a = e0 / 9, b = e0 - 9 * a;
      double m = ad0;
#pragma unroll
      for (int k = 0; k < 9; ++k) m -= sSub[9 * a + k] * sSub[9 * b + k];
      sM[e0] = m;
      if (has1) {
        const int a1 = e1 / 9, b1 = e1 - 9 * a1;
        double m1 = ad1;
#pragma unroll
        for (int k = 0; k < 9; ++k) m1 -= sSub[9 * a1 + k] * sSub[9 * b1 + k];
        sM[e1] = m1;
      }
    }
    __syncthreads();
    // lower Cholesky, row r = lane (lanes >= 9 carry zeros)
    {
      const int r = lane < 9 ? lane : 8;
      double x[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) x[c] = (lane < 9 && c <= r) ? sM[9 * r + c] : 0.0;
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        double d = rdlane64(x[c], c);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double inv = rsqrt(d);
        x[c] = (lane == c) ? d * inv : x[c] * inv;
#pragma unroll
        for (int cc = c + 1; cc < 9; ++cc) x[cc] -= x[c] * rdlane64(x[c], cc);  // garbage above the diagonal is never read
      }
      if (bad && lane == 0) atomicOr(P.flag, 1);
      __syncthreads();  // everyone has read sM
      if (lane < 9) {
#pragma unroll
        for (int c = 0; c < 9; ++c) sM[9 * lane + c] = (c <= lane) ? x[c] : 0.0;
      }
    }
    __syncthreads();
    // X = L^-1 (lower): lane = column; and the bracket of z
    if (lane < 9) {
      const int c = lane;
      double x[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) sum += (k >= c) ? sM[9 * r + k] * x[k] : 0.0;
        x[r] = (r == c) ? 1.0 / sM[10 * r] : (r > c ? -sum / sM[10 * r] : 0.0);
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) sX[9 * r + c] = x[r];
      double v = bs;
#pragma unroll
      for (int k = 0; k < 9; ++k) v -= sSub[9 * lane + k] * sz[k];
      sv[lane] = v;
    }
    __syncthreads();
    // publish the factor blocks, z, and the next sub-diagonal block L_{pos+1,pos} = Ae_{pos+1} L_kk^-T
    P.Ld[(size_t)81 * pos + e0] = sM[e0]; P.Ldinv[(size_t)81 * pos + e0] = sX[e0]; P.Lsub[(size_t)81 * pos + e0] = sSub[e0];
    if (has1) { P.Ld[(size_t)81 * pos + e1] = sM[e1]; P.Ldinv[(size_t)81 * pos + e1] = sX[e1]; P.Lsub[(size_t)81 * pos + e1] = sSub[e1]; }
    double znew = 0.0;
    if (lane < 9) {
      for (int k = 0; k <= lane; ++k) znew += sX[9 * lane + k] * sv[k];
      P.zs[(size_t)9 * pos + lane] = znew;
    }
    // Ae rows are needed in full by every entry: route them through LDS (sv is free again after the sync below)
    __shared__ double sAe[81];
    sAe[e0] = ae0;
    if (has1) sAe[e1] = ae1;
    __syncthreads();
    double nx0 = 0.0, nx1 = 0.0;
    if (pos + 1 < p1) {
      {
        const int a = e0 / 9, b = e0 - 9 * a;
#pragma unroll
        for (int c = 0; c < 9; ++c) nx0 += sAe[9 * a + c] * sX[9 * b + c];
      }
      if (has1) {
        const int a = e1 / 9, b = e1 - 9 * a;
#pragma unroll
        for (int c = 0; c < 9; ++c) nx1 += sAe[9 * a + c] * sX[9 * b + c];
      }
    }
    __syncthreads();
    sSub[e0] = nx0;
    if (has1) sSub[e1] = nx1;
    if (lane < 9) sz[lane] = znew;
    ad0 = nad0; ad1 = nad1; ae0 = nae0; ae1 = nae1; bs = nbs;
    __syncthreads();
  }
}

// Columns of Y = L_A^-1 B: one thread per pose dimension (column), marching down its chain from the first
// keyframe whose speed-bias block touches it. Y is K-major, so a workgroup's 256 columns are written coalesced.
// Every workgroup lies inside ONE chain and all its threads visit the same chain position in the same iteration
// (columns that start further down idle until the march reaches them): the position's factor blocks Lsub | Ldinv
// are then staged ONCE per workgroup in (double-buffered) LDS and read as broadcasts. The first version let every
// thread walk from its own start — 11 different positions per wave, 126 divergent global loads per step: 1.76 ms
// on the 5-agent map, all of it load latency.
__global__ __launch_bounds__(256) void k_sb_chain_cols(DevProblem P) {
  __shared__ __attribute__((aligned(16))) double sL[2][168];  // [buffer][Lsub 81 | pad 3 | Ldinv 81 | pad 3]
  const int tid = threadIdx.x;
  int b = blockIdx.x, c = 0;
  for (; c < P.nchains; ++c) {  // workgroup -> (chain, block of 256 columns inside it)
    const int nb = (6 * (P.chain_ptr[c + 1] - P.chain_ptr[c]) + 255) / 256;
    if (b < nb) break;
    b -= nb;
  }
  if (c >= P.nchains) return;
  const int p0 = P.chain_ptr[c], p1 = P.chain_ptr[c + 1];
  const int col = 6 * p0 + b * 256 + tid;
  const bool live = col < 6 * p1;
  const int q = live ? col / 6 : p1 - 1, e = col - 6 * q;  // chain position of the pose, component
  const int pstart = q > p0 ? q - 1 : q;                    // previous position if it is on the same chain
  const int pfirst = max((6 * p0 + b * 256) / 6 - 1, p0);   // first position any column of this workgroup needs
  auto fetch = [&](int pos) -> double {
    if (tid < 81) return P.Lsub[(size_t)81 * pos + tid];
    if (tid >= 84 && tid < 165) return P.Ldinv[(size_t)81 * pos + tid - 84];
    return 0.0;
  };
  double y[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) y[a] = 0.0;
  double nxt = fetch(pfirst);
  for (int pos = pfirst; pos < p1; ++pos) {
    double (&L)[168] = sL[(pos - pfirst) & 1];
    if (tid < 168) L[tid] = nxt;
    if (pos + 1 < p1) nxt = fetch(pos + 1);  // in flight during this step's arithmetic
    __syncthreads();                          // one barrier per step: the other buffer is rewritten two steps later
    if (!live || pos < pstart) continue;
    double v[9];
    const double* Bblk = (pos == q - 1) ? P.Bn : (pos == q ? P.Bs : (pos == q + 1 ? P.Bp : nullptr));
#pragma unroll
    for (int a = 0; a < 9; ++a) v[a] = Bblk ? Bblk[(size_t)54 * pos + 6 * a + e] : 0.0;
    if (pos > pstart) {
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int k = 0; k < 9; ++k) v[a] -= L[9 * a + k] * y[k];
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k <= a; ++k) s2 += L[84 + 9 * a + k] * v[k];
      y[a] = s2;
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) P.Y[(size_t)(9 * pos + a) * P.npad + col] = y[a];
  }
}

// b'_p[col] = b_p[col] - sum_k Y[k][col] z[k] over the column's chain segment. 8 lanes share a column (interleaved
// k), combined by a fixed-order butterfly: deterministic, and 8x more loads in flight than one thread per column.
__global__ __launch_bounds__(256) void k_pose_rhs(DevProblem P) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int col = gid >> 3, part = gid & 7;
  double acc = 0.0;
  if (col < 6 * P.K) {
    const int q = col / 6;
    const int kb = 9 * max(q - 1, 0), ke = 9 * P.pos_chain_end[q];
    for (int k = kb + part; k < ke; k += 8) acc += P.Y[(size_t)k * P.npad + col] * P.zs[k];
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
  if (part == 0 && col < 6 * P.K) P.bp[col] -= acc;
}

// x_s = A^-1 (b_s - B x_p): w = b_s - B x_p, forward with (Ldinv, Lsub), backward with their transposes.
// One 64-lane workgroup per chain; lanes 0..8 each own one row of the current 9-vector. Every step's operands are
// loaded one step ahead (the chain is latency-bound: ~900 dependent steps of 9x9 work per agent).
struct SbRowF { double w, ls[9], li[9]; };  // forward operands of one lane: bracket b_s - B x_p, Lsub row, Ldinv row
COV_DEV SbRowF sb_load_fwd(const DevProblem& P, int pos, int p0, int p1, int lane) {
  SbRowF o;
  double w = P.xs[(size_t)9 * pos + lane];
  const double* bs = P.Bs + (size_t)54 * pos + 6 * lane;
#pragma unroll
  for (int e = 0; e < 6; ++e) w -= bs[e] * P.bp[6 * pos + e];
  if (pos > p0) {
    const double* bpv = P.Bp + (size_t)54 * pos + 6 * lane;
#pragma unroll
    for (int e = 0; e < 6; ++e) w -= bpv[e] * P.bp[6 * (pos - 1) + e];
  }
  if (pos + 1 < p1) {
    const double* bn = P.Bn + (size_t)54 * pos + 6 * lane;
#pragma unroll
    for (int e = 0; e < 6; ++e) w -= bn[e] * P.bp[6 * (pos + 1) + e];
  }
  o.w = w;
#pragma unroll
  for (int k = 0; k < 9; ++k) { o.ls[k] = P.Lsub[(size_t)81 * pos + 9 * lane + k]; o.li[k] = P.Ldinv[(size_t)81 * pos + 9 * lane + k]; }
  return o;
}
struct SbRowB { double u, lsT[9], liT[9]; };  // backward operands: u_pos, column `lane` of Lsub_{pos+1} and of Ldinv_pos
COV_DEV SbRowB sb_load_bwd(const DevProblem& P, int pos, int p1, int lane) {
  SbRowB o;
  o.u = P.xs[(size_t)9 * pos + lane];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    o.lsT[k] = (pos + 1 < p1) ? P.Lsub[(size_t)81 * (pos + 1) + 9 * k + lane] : 0.0;
    o.liT[k] = P.Ldinv[(size_t)81 * pos + 9 * k + lane];
  }
  return o;
}

__global__ __launch_bounds__(64) void k_sb_backsolve(DevProblem P) {
  // the 9-vector of the previous step lives in lanes 0..8 and is broadcast with v_readlane (through SGPRs): no LDS
  // round trip and no barrier on the ~900-step dependent chain (the LDS version took 1.4 us per step)
  const int lane = threadIdx.x < 9 ? threadIdx.x : 8;
  const bool act = threadIdx.x < 9;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  double prev = 0.0;
  SbRowF cf = sb_load_fwd(P, p0, p0, p1, lane);
  for (int pos = p0; pos < p1; ++pos) {  // forward: u_pos = Linv (w_pos - Lsub u_{pos-1}), stored in xs
    SbRowF nf = cf;
    if (pos + 1 < p1) nf = sb_load_fwd(P, pos + 1, p0, p1, lane);
    double w = cf.w;
#pragma unroll
    for (int k = 0; k < 9; ++k) w -= cf.ls[k] * rdlane64(prev, k);  // Lsub of a chain head is zero
    double u = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) u += cf.li[k] * rdlane64(w, k);     // Ldinv is lower triangular: entries k > lane are zero
    if (act) P.xs[(size_t)9 * pos + lane] = u;
    prev = u;
    cf = nf;
  }
  prev = 0.0;
  __syncthreads();  // xs written above is re-read below by the same lanes; keeps the two sweeps ordered
  SbRowB cb = sb_load_bwd(P, p1 - 1, p1, lane);
  for (int pos = p1 - 1; pos >= p0; --pos) {  // backward: x_pos = Linv^T (u_pos - Lsub_{pos+1}^T x_{pos+1})
    SbRowB nb = cb;
    if (pos > p0) nb = sb_load_bwd(P, pos - 1, p1, lane);
    double w = cb.u;
#pragma unroll
    for (int k = 0; k < 9; ++k) w -= cb.lsT[k] * rdlane64(prev, k);
    double x = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) x += cb.liT[k] * rdlane64(w, k);    // column of a lower-triangular matrix: entries k < lane are zero
    if (act) P.xs[(size_t)9 * pos + lane] = x;
    prev = x;
    cb = nb;
  }
}

void launch_sb_chain_factor_early(const DevProblem& P, hipStream_t st, CholAux& ax) {
  if (!P.vi) return;
  ax.init();
  hipLaunchKernelGGL(k_gather_rhs, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, 1);
  (void)hipEventRecord(ax.ev_sb, st);
  (void)hipStreamWaitEvent(ax.aux, ax.ev_sb, 0);
  hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, ax.aux, P);
  (void)hipEventRecord(ax.ev_cf, ax.aux);
  ax.cf_pending = true;
}

void launch_structured_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax) {
  const int cnt = P.npad > 9 * P.K ? P.npad : 9 * P.K;
  const bool early = P.vi && ax.cf_pending;  // chain factor (and z in xs) already under way on the auxiliary stream
  hipLaunchKernelGGL(k_gather_rhs, dim3((cnt + 255) / 256), dim3(256), 0, st, P, early ? 0 : 2);
  if (P.vi) {
    if (early) { (void)hipStreamWaitEvent(st, ax.ev_cf, 0); ax.cf_pending = false; }
    else hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, st, P);
    hipLaunchKernelGGL(k_sb_chain_cols, dim3((6 * P.K + 255) / 256 + P.nchains), dim3(256), 0, st, P);  // >= sum of per-chain block counts
    hipLaunchKernelGGL(k_pose_rhs, dim3((8 * 6 * P.K + 255) / 256), dim3(256), 0, st, P);
    launch_yty_update(P, st);
  }
  dense_cholesky_solve_raw(P.Sred, P.bp, P.Linv, P.flag, P.npad, st, ax);
  if (P.vi) hipLaunchKernelGGL(k_sb_backsolve, dim3(P.nchains), dim3(64), 0, st, P);
  hipLaunchKernelGGL(k_scatter_solution, dim3((P.n + 255) / 256), dim3(256), 0, st, P, dst);
}

}  // namespace covgpu
