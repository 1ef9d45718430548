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
//   sb_chain_factor   one wave per IMU chain, sequential over keyframes, the factor only: L_kk L_kk^T = Ad_k - Lsub Lsub^T
//                     and L_kk^-1 in one sweep, Lsub_{k+1} = Ae_{k+1} L_kk^-T (operands staged through LDS a chunk ahead)
//   sb_sweep<+1|-1>   the two LINEAR recurrences along a chain as one 9x9 product per step: z = L_A^-1 b_s (forward) and
//                     x = L_A^-T u (backward); their matrices depend on the factor only (sb_sweep_mat), the vectors are
//                     parallel products (sb_sweep_vec)
//   sb_chain_cols     Y = L_A^-1 B on the matrix core, one wave per 16-column tile marching down its chain; past the
//                     injection rows with the propagator M_pos = -Ldinv_pos Lsub_pos (one product per position)
//   pose_rhs_sweep    b'_p = b_p - B^T A^-1 b_s from the sweeps (no pass over Y)
//   sb_gram, yty_semisep   C -= Y^T Y through the semiseparable structure (O(K^2), one pass over Y)
//   (k_chol.hip, k_panel.hip, k_arrow.hip)   dense / block-arrow Cholesky of C'
//   sb_fwd_matvec + sb_sweep<-1>   x_s = A^-1 (b_s - B x_p): u = z - Y x_p in parallel, then the backward sweep
#include <cstdlib>

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
// Single-wave workgroups: LDS operations of one wave execute in order, so ordering LDS traffic needs no s_barrier — and
// above all not __syncthreads(), whose s_waitcnt vmcnt(0) also waits for every outstanding GLOBAL store to be acknowledged
// (~1.5 us per chain step here: the factor blocks stream out while the recurrence goes on, nobody reads them back).
COV_DEV void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
COV_DEV void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kCfChunk = 8;                     // positions staged per LDS buffer
constexpr int kCfPerLane = (kCfChunk * 81 + 63) / 64;
// One wave per IMU chain, sequential over keyframes — the block-bidiagonal Cholesky factor only (z = L_A^-1 b_s is a LINEAR
// recurrence and runs separately, k_sb_sweep). Per position:
//   S  = Ad - W W^T          (W = Lsub_pos; lower triangle only: 45 entries, one per lane)
//   L  = chol(S), X = L^-1   in ONE sweep: lanes 0..8 hold the rows of S, lanes 16..24 the columns of an identity; every
//                            pivot scales register c by 1/sqrt(d) and subtracts x[c] * L[cc][c] (broadcast from lane cc by
//                            v_readlane) from register cc — for the first group that is the right-looking Cholesky, for
//                            the second the column-oriented forward substitution L X = I with the very same multipliers:
//                            the inverse costs no instruction of its own
//   W' = Ae_{pos+1} X^T      (81 entries, two rounds)
// Inputs are staged a chunk ahead as plain strided copies of contiguous ranges, results leave through LDS with one coalesced
// store per chunk (a gather with per-element index arithmetic, and four global stores per step, cost more instructions than
// the recurrence itself: 3.2 us per position before, measured alone).
__global__ __launch_bounds__(64) void k_sb_chain_factor(DevProblem P) {
  __shared__ __attribute__((aligned(16))) double sW[9 * 10], sS[9 * 10], sX[9 * 10];   // pitch 10: rows are 16-byte aligned
  __shared__ double sAd[2][kCfChunk * 81], sAe[2][kCfChunk * 81];
  __shared__ double sOutL[kCfChunk * 81], sOutS[kCfChunk * 81];
  const int lane = threadIdx.x;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const int nchunk = (p1 - p0 + kCfChunk - 1) / kCfChunk;
  // lower-triangle entry of this lane: (ta, tb), tb <= ta (lanes >= 45 idle in that phase)
  int ta = 0, tb = 0;
  { int t = lane < 45 ? lane : 44; while (t > ta) { t -= ta + 1; ++ta; } tb = t; }
  const int e0 = lane, e1 = lane + 64;
  const int a0 = e0 / 9, b0 = e0 - 9 * a0, a1 = (e1 < 81 ? e1 : 80) / 9, b1 = (e1 < 81 ? e1 : 80) - 9 * a1;
  for (int e = lane; e < 90; e += 64) { sW[e] = 0.0; sX[e] = 0.0; sS[e] = 0.0; }
  double stA[kCfPerLane], stE[kCfPerLane];
  auto gload = [&](int c) {
    const int base = p0 + c * kCfChunk, len = min(kCfChunk, p1 - base);
    const double* srcA = P.Ad + (size_t)81 * base;
    const double* srcE = P.Ae + (size_t)81 * (base + 1);     // Ae of the NEXT position; none behind the chain's last
    const int cntA = 81 * len, cntE = 81 * min(len, p1 - base - 1);
#pragma unroll
    for (int i = 0; i < kCfPerLane; ++i) {
      const int idx = lane + 64 * i;
      stA[i] = (idx < cntA) ? srcA[idx] : 0.0;
      stE[i] = (idx < cntE) ? srcE[idx] : 0.0;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kCfPerLane; ++i) {
      const int idx = lane + 64 * i;
      if (idx < kCfChunk * 81) { sAd[buf][idx] = stA[i]; sAe[buf][idx] = stE[i]; }
    }
  };
  bool bad = false;
  gload(0);
  for (int c = 0; c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1);
    wave_sync();
    const int base = p0 + c * kCfChunk, len = min(kCfChunk, p1 - base);
    for (int q = 0; q < len; ++q) {
      const double* Ad = sAd[c & 1] + 81 * q;
      const double* Ae = sAe[c & 1] + 81 * q;
      // ---- S = Ad - W W^T, lower triangle
      {
        const double* wa = sW + 10 * ta;
        const double* wb = sW + 10 * tb;
        double m0 = Ad[9 * ta + tb], m1 = 0.0, m2 = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k += 3) { m0 -= wa[k] * wb[k]; m1 -= wa[k + 1] * wb[k + 1]; m2 -= wa[k + 2] * wb[k + 2]; }
        if (lane < 45) sS[10 * ta + tb] = (m0 + m1) + m2;
        // Lsub_pos leaves now (it is overwritten below)
        sOutS[81 * q + e0] = sW[10 * a0 + b0];
        if (e1 < 81) sOutS[81 * q + e1] = sW[10 * a1 + b1];
      }
      wave_sync();
      // ---- Cholesky and inverse in one sweep
      {
        const int g = lane >> 4, r = lane & 15;
        double x[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double sv = sS[10 * (r < 9 ? r : 8) + k];
          x[k] = (g == 0) ? ((r < 9 && k <= r) ? sv : 0.0) : ((g == 1 && k == r) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          double d = rdlane64(x[k], k);
          if (!(d > 0.0)) { bad = true; d = 1.0; }
          double rs = __builtin_amdgcn_rsq(d);
          rs = rs * (1.5 - 0.5 * d * rs * rs);
          rs = rs * (1.5 - 0.5 * d * rs * rs);
          x[k] *= rs;  // (lane k of the first group holds d itself: d * rs = sqrt(d))
#pragma unroll
          for (int cc = k + 1; cc < 9; ++cc) x[cc] -= x[k] * rdlane64(x[k], cc);
        }
        if (g == 1 && r < 9) {  // column r of X = L^-1: X[k][r] = x[k] (zero above the diagonal)
#pragma unroll
          for (int k = 0; k < 9; ++k) { sX[10 * k + r] = x[k]; sOutL[81 * q + 9 * k + r] = x[k]; }
        }
      }
      wave_sync();
      // ---- W' = Ae_{pos+1} X^T (zero behind the chain's last position: Ae staged as zeros)
      {
        const double* xa = sX + 10 * b0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k += 3) { s0 += Ae[9 * a0 + k] * xa[k]; s1 += Ae[9 * a0 + k + 1] * xa[k + 1]; s2 += Ae[9 * a0 + k + 2] * xa[k + 2]; }
        const double w0 = (s0 + s1) + s2;
        double w1 = 0.0;
        if (e1 < 81) {
          const double* xb = sX + 10 * b1;
          double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
          for (int k = 0; k < 9; k += 3) { t0 += Ae[9 * a1 + k] * xb[k]; t1 += Ae[9 * a1 + k + 1] * xb[k + 1]; t2 += Ae[9 * a1 + k + 2] * xb[k + 2]; }
          w1 = (t0 + t1) + t2;
        }
        wave_sync();  // every lane has read sW (above) and sX before they change
        sW[10 * a0 + b0] = w0;
        if (e1 < 81) sW[10 * a1 + b1] = w1;
      }
      wave_sync();
    }
    // ---- results of the chunk: contiguous ranges of Ldinv and Lsub
    for (int e = lane; e < 81 * len; e += 64) {
      P.Ldinv[(size_t)81 * base + e] = sOutL[e];
      P.Lsub[(size_t)81 * base + e] = sOutS[e];
    }
    wave_sync();
  }
  if (bad && lane == 0) atomicOr(P.flag, 1);
}

// ---- linear recurrences along a chain as ONE 9x9 matrix-vector product per step -------------------------------------
//   DIR = +1  z_pos = f_pos - T_pos z_{pos-1},  T_pos = Ldinv_pos Lsub_pos, f_pos = Ldinv_pos b_pos          (z = L_A^-1 b_s)
//   DIR = -1  x_pos = v_pos - N_pos^T x_{pos+1}, N_pos = Lsub_{pos+1} Ldinv_pos, v_pos = Ldinv_pos^T u_pos   (x_s = L_A^-T u)
// T / N depend on the factor only (k_sb_sweep_mat, auxiliary stream), f / v are parallel products (k_sb_sweep_vec). The
// operands of a position are 90 consecutive doubles [9x9 matrix, row i = the coefficients of output i | 9-vector], so a
// chunk of positions is one contiguous range and its staging a plain strided copy. The two-product form it replaces
// (k_sb_backsolve) took 0.58 us per position, this one 0.21 (tools/chain_probe.hip).
constexpr int kBkChunk = 16;      // positions staged per LDS buffer
constexpr int kBkBlk = 90;        // per position: matrix 81 | vector 9
constexpr int kBkPerLane = (kBkChunk * kBkBlk + 63) / 64;
__global__ __launch_bounds__(256) void k_sb_sweep_mat(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 81 * P.K) return;
  const int pos = q / 81, e = q - 81 * pos, i = e / 9, k = e - 9 * i;
  {  // forward: T[i][k] = sum_{c <= i} Ldinv_pos[i][c] Lsub_pos[c][k]
    const double* A = P.Ldinv + (size_t)81 * pos + 9 * i;
    const double* B = P.Lsub + (size_t)81 * pos + k;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) s += (c <= i) ? A[c] * B[9 * c] : 0.0;
    P.Zfwd[(size_t)kBkBlk * pos + e] = s;
  }
  {  // backward: N^T[i][k] = N[k][i] = sum_{c >= i} Lsub_{pos+1}[k][c] Ldinv_pos[c][i]
    double s = 0.0;
    if (pos + 1 < P.pos_chain_end[pos]) {
      const double* A = P.Lsub + (size_t)81 * (pos + 1) + 9 * k;
      const double* B = P.Ldinv + (size_t)81 * pos + i;
#pragma unroll
      for (int c = 0; c < 9; ++c) s += (c >= i) ? A[c] * B[9 * c] : 0.0;
    }
    P.Nback[(size_t)kBkBlk * pos + e] = s;
  }
}
template <int DIR>
__global__ __launch_bounds__(256) void k_sb_sweep_vec(DevProblem P, const double* __restrict__ src) {  // DIR +1: f = Ldinv src -> Zfwd | -1: v = Ldinv^T src -> Nback
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 9 * P.K) return;
  const int pos = q / 9, i = q - 9 * pos;
  const double* L = P.Ldinv + (size_t)81 * pos;
  const double* u = src + (size_t)9 * pos;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) s += (DIR > 0) ? ((k <= i) ? L[9 * i + k] * u[k] : 0.0) : ((k >= i) ? L[9 * k + i] * u[k] : 0.0);
  (DIR > 0 ? P.Zfwd : P.Nback)[(size_t)kBkBlk * pos + 81 + i] = s;
}
template <int DIR>
__global__ __launch_bounds__(64) void k_sb_sweep(DevProblem P) {
  __shared__ double sblk[2][kBkChunk * kBkBlk];
  __shared__ double sOut[kBkChunk * 9];
  const int tid = threadIdx.x;
  const int lane = tid < 9 ? tid : 8;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const int nchunk = (p1 - p0 + kBkChunk - 1) / kBkChunk;
  const double* arr = DIR > 0 ? P.Zfwd : P.Nback;
  double* out = DIR > 0 ? P.zs : P.xs;
  // chunk c: DIR +1 positions [p0 + c C, ...), DIR -1 positions [max(p1 - (c+1) C, p0), p1 - c C); slot q = pos - base
  auto cbase = [&](int c) { return DIR > 0 ? p0 + c * kBkChunk : max(p1 - (c + 1) * kBkChunk, p0); };
  auto clen = [&](int c) { return DIR > 0 ? min(kBkChunk, p1 - (p0 + c * kBkChunk)) : p1 - c * kBkChunk - max(p1 - (c + 1) * kBkChunk, p0); };
  double stage[kBkPerLane];
  auto gload = [&](int c) {
    const int cnt = clen(c) * kBkBlk;
    const double* src = arr + (size_t)kBkBlk * cbase(c);
#pragma unroll
    for (int i = 0; i < kBkPerLane; ++i) { const int idx = tid + 64 * i; stage[i] = (idx < cnt) ? src[idx] : 0.0; }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kBkPerLane; ++i) { const int idx = tid + 64 * i; if (idx < kBkChunk * kBkBlk) sblk[buf][idx] = stage[i]; }
  };
  double x = 0.0;
  gload(0);
  for (int c = 0; c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1);
    wave_sync();
    const int base = cbase(c), len = clen(c);
    const int qfirst = DIR > 0 ? 0 : len - 1;
    double row[9], v;  // operands of the CURRENT step, fetched one step ahead of the dependent chain
    {
      const double* L = sblk[c & 1] + kBkBlk * qfirst;
#pragma unroll
      for (int k = 0; k < 9; ++k) row[k] = L[9 * lane + k];
      v = L[81 + lane];
    }
    for (int s = 0; s < len; ++s) {
      const int q = DIR > 0 ? s : len - 1 - s;
      const int qn = (s + 1 < len) ? q + DIR : q;
      double nrow[9], nv;
      const double* Ln = sblk[c & 1] + kBkBlk * qn;
#pragma unroll
      for (int k = 0; k < 9; ++k) nrow[k] = Ln[9 * lane + k];
      nv = Ln[81 + lane];
      double b[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) b[k] = rdlane64(x, k);  // all nine broadcasts first, then three short accumulation chains
      const double s0 = fma(row[0], b[0], fma(row[3], b[3], row[6] * b[6]));
      const double s1 = fma(row[1], b[1], fma(row[4], b[4], row[7] * b[7]));
      const double s2 = fma(row[2], b[2], fma(row[5], b[5], row[8] * b[8]));
      x = v - ((s0 + s1) + s2);
      if (tid < 9) sOut[9 * q + tid] = x;  // (results leave through LDS, one coalesced store per chunk)
#pragma unroll
      for (int k = 0; k < 9; ++k) row[k] = nrow[k];
      v = nv;
    }
    wave_sync();
    for (int e = tid; e < 9 * len; e += 64) out[(size_t)9 * base + e] = sOut[e];
    wave_sync();
  }
}

// Columns of Y = L_A^-1 B, marching down the chain: at position pos
//     T = B_inj - Lsub_pos Y(pos-1)          Y(pos) = Ldinv_pos T
// for all columns at once — two (9x9)x(9x16) products per 16-column tile, done on the matrix core by ONE WAVE PER
// TILE with no LDS and no barrier: for v_mfma_f64_16x16x4 the C/D register layout (row = (lane>>4) + 4 reg,
// col = lane & 15) is exactly the B-operand layout of k-step `reg` (k = (lane>>4) + 4 reg, n = lane & 15), so the
// accumulator of one product is fed straight back as the B operand of the next. The 9x9 blocks are zero-padded to
// 16x12 (A operands, 3 + 3 doubles per lane per position, fetched four positions ahead); the B injection (pose j
// enters at positions j-1, j, j+1 through Bn, Bs, Bp) is the accumulator's initial value; columns whose pose lies
// further down stay exactly zero until the march reaches them. Y is K-major: a register row is stored as 16
// consecutive doubles. (Scalar versions: one thread per column from its own start 1.76 ms; lockstep positions with
// LDS-staged blocks 0.80 ms; this one is bound by six dependent MFMAs per position.)
typedef double v4f64s __attribute__((ext_vector_type(4)));
struct ColOps { double ls[3], li[3], bi[3]; };
__global__ __launch_bounds__(256) void k_sb_chain_cols(DevProblem P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup -> (chain, block of 64 columns inside it): host-built list (a sharded rank holds thousands of one-keyframe
  // chains — the keyframes of other ranks' agents — and a linear search over the chains per workgroup cost 0.7 ms each)
  const int c = P.cc_chain[blockIdx.x], b = P.cc_blk[blockIdx.x];
  const int p0 = P.chain_ptr[c], p1 = P.chain_ptr[c + 1];
  const int col0 = 6 * p0 + (b * 4 + wave) * 16;
  if (col0 >= 6 * p1) return;  // wave-uniform; the kernel has no barrier
  const int n = lane & 15, kq = lane >> 4;
  const int col = col0 + n;
  const bool live = col < 6 * p1;
  const int q = live ? col / 6 : p1 - 1, e = live ? col - 6 * q : 0;  // chain position of the pose, component
  const int pstart = q > p0 ? q - 1 : q;
  const int pfirst = max(col0 / 6 - 1, p0);
  const size_t ld = (size_t)P.Yld[c];
  double* const Yc = P.Y + P.Yoff[c];  // this chain's block: rows 9 (pos - p0), columns col - 6 p0
  auto fetch = [&](int pos) {
    ColOps o;
    const bool in = pos < p1;
    const double* Bblk = (!live || !in) ? nullptr : (pos == q - 1 ? P.Bn : (pos == q ? P.Bs : (pos == q + 1 ? P.Bp : nullptr)));
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int k = kq + 4 * s2;                 // A operand: A[m = n][k]; injection / C rows: row = k
      const bool ok = in && n < 9 && k < 9;
      o.ls[s2] = ok ? -P.Lsub[(size_t)81 * pos + 9 * n + k] : 0.0;
      o.li[s2] = (ok && k <= n) ? P.Ldinv[(size_t)81 * pos + 9 * n + k] : 0.0;
      o.bi[s2] = (Bblk != nullptr && k < 9) ? Bblk[(size_t)54 * pos + 6 * k + e] : 0.0;
    }
    return o;
  };
  // Head: the positions where some column of the tile is still being injected (pose q enters at q-1, q, q+1) — the full
  // two-product form. Tail: every column obeys y(pos) = M_pos y(pos-1) with the propagator M_pos = -Ldinv_pos Lsub_pos
  // (k_sb_propagator, right behind the chain factorisation): ONE product per position, three operand loads instead of
  // nine, fetched eight positions ahead through a ring of registers (loop unrolled by the ring size: no copies).
  const int ptail = min((col0 + 15) / 6 + 2, p1);  // first position at which no column of the tile is injected any more
  ColOps o0 = fetch(pfirst), o1 = fetch(pfirst + 1), o2 = fetch(pfirst + 2), o3 = fetch(pfirst + 3);
  v4f64s y = {0.0, 0.0, 0.0, 0.0};
  auto store = [&](int pos) {
    if (live && pos >= pstart) {
      double* Yp = Yc + (size_t)(9 * (pos - p0) + kq) * ld + (col - 6 * p0);
      Yp[0] = y[0];
      Yp[4 * ld] = y[1];
      if (kq == 0) Yp[8 * ld] = y[2];
    }
  };
  for (int pos = pfirst; pos < ptail; ++pos) {
    const ColOps o = o0;
    o0 = o1; o1 = o2; o2 = o3; o3 = fetch(pos + 4);
    v4f64s t = {o.bi[0], o.bi[1], o.bi[2], 0.0};
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ls[0], y[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ls[1], y[1], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ls[2], y[2], t, 0, 0, 0);
    v4f64s yn = {0.0, 0.0, 0.0, 0.0};
    yn = __builtin_amdgcn_mfma_f64_16x16x4f64(o.li[0], t[0], yn, 0, 0, 0);
    yn = __builtin_amdgcn_mfma_f64_16x16x4f64(o.li[1], t[1], yn, 0, 0, 0);
    yn = __builtin_amdgcn_mfma_f64_16x16x4f64(o.li[2], t[2], yn, 0, 0, 0);
    y = yn;
    store(pos);
  }
  constexpr int RING = 8;
  double mr[RING][3];
  auto mfetch = [&](int pos, double* m) {
    const bool in = pos < p1;
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int k = kq + 4 * s2;
      m[s2] = (in && n < 9 && k < 9) ? P.Mblk[(size_t)81 * pos + 9 * n + k] : 0.0;
    }
  };
#pragma unroll
  for (int i = 0; i < RING; ++i) mfetch(ptail + i, mr[i]);
  for (int pos = ptail; pos < p1; pos += RING) {
#pragma unroll
    for (int i = 0; i < RING; ++i) {
      if (pos + i < p1) {  // wave-uniform
        v4f64s yn = {0.0, 0.0, 0.0, 0.0};
        yn = __builtin_amdgcn_mfma_f64_16x16x4f64(mr[i][0], y[0], yn, 0, 0, 0);
        yn = __builtin_amdgcn_mfma_f64_16x16x4f64(mr[i][1], y[1], yn, 0, 0, 0);
        yn = __builtin_amdgcn_mfma_f64_16x16x4f64(mr[i][2], y[2], yn, 0, 0, 0);
        y = yn;
        mfetch(pos + i + RING, mr[i]);
        store(pos + i);
      }
    }
  }
}

// ---- C -= Y^T Y without the cubic sum (semiseparable structure) ------------------------------------------------
// Below its injection rows a column of Y obeys a linear recurrence: y(pos) = M_pos y(pos-1), M_pos = -Ldinv_pos Lsub_pos
// (pose j enters through B at positions j-1, j, j+1 only). For poses i <= j of one chain therefore
//   (Y^T Y)_ij = y_i(j-1)^T y_j(j-1) + y_i(j)^T y_j(j) + y_i(j+1)^T (I + G_{j+1}) y_j(j+1),
//   G_q = sum_{pos > q} Phi(pos,q)^T Phi(pos,q) = M_{q+1}^T (I + G_{q+1}) M_{q+1},   G_{last} = 0,
// i.e. pose j's six rows of C need only the THREE 9-row groups j-1, j, j+1 of Y: O(K^2) work and one pass over Y
// instead of the O(K^3) MFMA product over whole chain segments (76 GFLOP, 1.9 ms on the 5-agent map).
__global__ __launch_bounds__(256) void k_sb_propagator(DevProblem P) {  // Mblk[pos] = -Ldinv_pos Lsub_pos
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 81 * P.K) return;
  const int pos = t / 81, e = t - 81 * pos, a = e / 9, b = e - 9 * a;
  const double* Li = P.Ldinv + (size_t)81 * pos + 9 * a;
  const double* Ls = P.Lsub + (size_t)81 * pos + b;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc += (k <= a) ? Li[k] * Ls[9 * k] : 0.0;
  P.Mblk[t] = -acc;
}

// GI[q] = I + G_q by the backward recurrence, one workgroup per chain (81 active threads, one matrix entry each).
__global__ __launch_bounds__(128) void k_sb_gram(DevProblem P) {
  __shared__ double sG[81], sM[2][81], sT[81];
  const int tid = threadIdx.x;
  const bool act = tid < 81;
  const int a = act ? tid / 9 : 0, b = act ? tid - 9 * a : 0;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const double eye = (a == b) ? 1.0 : 0.0;
  if (act) { sG[tid] = eye; P.GI[(size_t)81 * (p1 - 1) + tid] = eye; }
  auto mload = [&](int pos) { return (act && pos > p0 && pos < p1) ? P.Mblk[(size_t)81 * pos + tid] : 0.0; };
  double m0 = mload(p1 - 1), m1 = mload(p1 - 2), m2 = mload(p1 - 3), m3 = mload(p1 - 4);  // four steps of loads in flight
  int it = 0;
  for (int q = p1 - 2; q >= p0; --q, ++it) {  // uses M_{q+1}
    double (&M)[81] = sM[it & 1];
    if (act) M[tid] = m0;
    m0 = m1; m1 = m2; m2 = m3; m3 = mload(q - 3);
    lds_barrier2();  // LDS traffic only: the GI stores of the previous step need not be acknowledged
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) t += sG[9 * a + k] * M[9 * k + b];   // T = (I + G_{q+1}) M
    if (act) sT[tid] = t;
    lds_barrier2();  // LDS traffic only: the GI stores of the previous step need not be acknowledged
    double g = eye;
#pragma unroll
    for (int k = 0; k < 9; ++k) g += M[9 * k + a] * sT[9 * k + b];   // I + M^T T
    if (act) { sG[tid] = g; P.GI[(size_t)81 * q + tid] = g; }
    // (the next iteration writes the OTHER sM buffer; its first barrier orders this sG write before the reads)
  }
}

// Workgroup (j, chunk): C[rows 6j..6j+5, 256 columns of the poses <= j in the chain] -= W_j^T Y[rows 9(j-1) .. 9(j+2), cols]
// with W_j = [y_j(j-1); y_j(j); (I + G_{j+1}) y_j(j+1)] (27 x 6, built in LDS first — every chunk rebuilds it, 162 short
// dot products against 256 x 27 loads). One workgroup per pose walked up to twelve column chunks for the last poses of a
// chain and one for the first: the launch ended with a long tail of a few busy workgroups.
__global__ __launch_bounds__(256) void k_yty_semisep(DevProblem P) {
  __shared__ double sW[27][6];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int p1 = P.pos_chain_end[j], p0 = P.pos_chain_begin[j], ch = P.pos_chain[j];
  const int c = 6 * p0 + 256 * (int)blockIdx.y + tid;
  if (c - tid >= 6 * (j + 1)) return;  // whole chunk right of the pose's own columns (workgroup-uniform)
  const size_t ld = (size_t)P.Yld[ch];
  const double* const Yc = P.Y + P.Yoff[ch] - (size_t)9 * p0 * ld - (size_t)6 * p0;  // addressed with GLOBAL (row, column) below
  if (tid < 162) {
    const int r = tid / 6, e = tid - 6 * r, grp = r / 9, a = r - 9 * grp, pos = j - 1 + grp;
    double v = 0.0;
    if (pos >= p0 && pos < p1) {
      if (grp < 2) v = Yc[(size_t)(9 * pos + a) * ld + 6 * j + e];
      else {
        const double* G = P.GI + (size_t)81 * pos + 9 * a;
#pragma unroll
        for (int k = 0; k < 9; ++k) v += G[k] * Yc[(size_t)(9 * pos + k) * ld + 6 * j + e];
      }
    }
    sW[r][e] = v;
  }
  __syncthreads();
  const int r0 = (j - 1 >= p0) ? 0 : 9, r1 = (j + 1 < p1) ? 27 : 18;  // row groups that exist in this chain
  if (c < 6 * (j + 1)) {
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int r = r0; r < r1; ++r) {
      const double y = Yc[(size_t)(9 * (j - 1) + r) * ld + c];
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[a] += sW[r][a] * y;
    }
    const int pc = c / 6, ec = c - 6 * pc;
#pragma unroll
    for (int a = 0; a < 6; ++a)
      if (c <= 6 * j + a) *c_entry(P, j, pc, a, ec) -= acc[a];
  }
}

// b'_p[col] = b_p[col] - sum_k Y[k][col] z[k] over the column's chain segment. 8 lanes share a column (interleaved
// k), combined by a fixed-order butterfly: deterministic, and 8x more loads in flight than one thread per column.
__global__ __launch_bounds__(256) void k_pose_rhs(DevProblem P) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int col = gid >> 3, part = gid & 7;
  double acc = 0.0;
  if (col < 6 * P.K) {
    const int q = col / 6, p0 = P.pos_chain_begin[q], ch = P.pos_chain[q];
    const size_t ld = (size_t)P.Yld[ch];
    const double* const Yc = P.Y + P.Yoff[ch] - (size_t)9 * p0 * ld - (size_t)6 * p0;
    const int kb = 9 * max(q - 1, p0), ke = 9 * P.pos_chain_end[q];
    for (int k = kb + part; k < ke; k += 8) acc += Yc[(size_t)k * ld + col] * P.zs[k];
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
  if (part == 0 && col < 6 * P.K) P.bp[col] -= acc;
}

// b'_p = b_p - B^T x0 with x0 = A^-1 b_s (in xs, left by the sweeps behind the chain factorisation): pose j couples to the
// speed-bias blocks of positions j (Bs), j+1 (through its Bp) and j-1 (through its Bn) of its chain. One thread per entry.
__global__ __launch_bounds__(256) void k_pose_rhs_sweep(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 6 * P.K) return;
  const int j = t / 6, e = t - 6 * j;
  const int pb = P.pos_chain_begin[j], pe = P.pos_chain_end[j];
  double acc = 0.0;
  {
    const double* B = P.Bs + (size_t)54 * j + e;
    const double* x = P.xs + (size_t)9 * j;
#pragma unroll
    for (int r = 0; r < 9; ++r) acc += B[6 * r] * x[r];
  }
  if (j + 1 < pe) {
    const double* B = P.Bp + (size_t)54 * (j + 1) + e;
    const double* x = P.xs + (size_t)9 * (j + 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) acc += B[6 * r] * x[r];
  }
  if (j > pb) {
    const double* B = P.Bn + (size_t)54 * (j - 1) + e;
    const double* x = P.xs + (size_t)9 * (j - 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) acc += B[6 * r] * x[r];
  }
  P.bp[t] -= acc;
}

// x_s = A^-1 (b_s - B x_p), in two kernels.
// k_sb_rhs (one thread per speed-bias row, fully parallel): w = b_s - B x_p, in place in xs.
__global__ __launch_bounds__(256) void k_sb_rhs(DevProblem P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9 * P.K) return;
  const int pos = t / 9, r = t - 9 * pos;
  const int p1 = P.pos_chain_end[pos];
  double w = P.xs[t];
  const double* bs = P.Bs + (size_t)54 * pos + 6 * r;
#pragma unroll
  for (int e = 0; e < 6; ++e) w -= bs[e] * P.bp[6 * pos + e];
  if (pos > 0 && P.pos_chain_end[pos - 1] == p1) {  // not the head of its chain
    const double* bpv = P.Bp + (size_t)54 * pos + 6 * r;
#pragma unroll
    for (int e = 0; e < 6; ++e) w -= bpv[e] * P.bp[6 * (pos - 1) + e];
  }
  if (pos + 1 < p1) {
    const double* bn = P.Bn + (size_t)54 * pos + 6 * r;
#pragma unroll
    for (int e = 0; e < 6; ++e) w -= bn[e] * P.bp[6 * (pos + 1) + e];
  }
  P.xs[t] = w;
}

// u = L_A^-1 (b_s - B x_p) WITHOUT a sequential sweep: L_A^-1 b_s = z (left by the chain factorisation) and L_A^-1 B = Y (stored
// for the Schur complement anyway), so u = z - Y x_p is one parallel pass over Y (430 MB on the 5-agent map) instead of the
// forward half of the 9x9 bidiagonal recursion (440 dependent steps per agent). One wave per row of Y; a row of chain
// position pos has non-zeros in the columns of poses 0 .. pos+1 of its chain only (trapezoid).
__global__ __launch_bounds__(256) void k_sb_fwd_matvec(DevProblem P) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 9 * P.K) return;
  const int pos = row / 9, a = row - 9 * pos, p0 = P.pos_chain_begin[pos], p1 = P.pos_chain_end[pos], ch = P.pos_chain[pos];
  const size_t ld = (size_t)P.Yld[ch];
  const double* y = P.Y + P.Yoff[ch] + (size_t)(9 * (pos - p0) + a) * ld;
  const int ncol = 6 * (min(pos + 2, p1) - p0);
  const double* xp = P.bp + 6 * p0;
  double acc = 0.0;
  for (int c = lane; c < ncol; c += 64) acc += y[c] * xp[c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) P.xs[row] = P.zs[row] - acc;
}

// k_sb_backsolve: forward with (Ldinv, Lsub), backward with their transposes — one wave per chain, ~900 dependent
// steps of 9x9 work per agent. Lanes 0..8 own one row of the current 9-vector; the previous step's vector is
// broadcast with v_readlane (no barrier on the chain). The factor blocks are staged through LDS in chunks of
// kSbChunk positions by all 64 lanes, the next chunk's global loads in flight while the current one is consumed:
// the first version loaded each step's operands one step ahead and ran at HBM latency (1.4 us per step).
constexpr int kSbChunk = 8;
constexpr int kSbBlk = 168;  // Lsub 81 | pad 3 | Ldinv 81 | pad 3
constexpr int kSbPerLane = (kSbChunk * 162 + 63) / 64;
__global__ __launch_bounds__(64) void k_sb_backsolve(DevProblem P, int do_forward) {
  __shared__ __attribute__((aligned(16))) double sblk[2][kSbChunk][kSbBlk];
  const int tid = threadIdx.x;
  const int lane = tid < 9 ? tid : 8;
  const bool act = tid < 9;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const int nchunk = (p1 - p0 + kSbChunk - 1) / kSbChunk;
  double stage[kSbPerLane];
  // chunk c covers positions [p0 + c*kSbChunk, ...) in ascending order (dir = +1) or the mirrored range (dir = -1)
  auto chunk_base = [&](int c, int dir) { return dir > 0 ? p0 + c * kSbChunk : max(p1 - (c + 1) * kSbChunk, p0); };
  auto chunk_len = [&](int c, int dir) { return dir > 0 ? min(kSbChunk, p1 - (p0 + c * kSbChunk)) : min(kSbChunk, p1 - c * kSbChunk - p0); };
  auto gload = [&](int c, int dir) {
    const int base = chunk_base(c, dir), len = chunk_len(c, dir);
#pragma unroll
    for (int i = 0; i < kSbPerLane; ++i) {
      const int idx = tid + 64 * i, q = idx / 162, e = idx - 162 * q;
      stage[i] = (q < len) ? (e < 81 ? P.Lsub[(size_t)81 * (base + q) + e] : P.Ldinv[(size_t)81 * (base + q) + e - 81]) : 0.0;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kSbPerLane; ++i) {
      const int idx = tid + 64 * i, q = idx / 162, e = idx - 162 * q;
      if (q < kSbChunk) sblk[buf][q][e < 81 ? e : e + 3] = stage[i];
    }
  };
  // ---- forward: u_pos = Linv (w_pos - Lsub u_{pos-1}), stored in xs (skipped when k_sb_fwd_matvec already left u in xs)
  double prev = 0.0;
  if (do_forward) gload(0, +1);
  for (int c = 0; do_forward && c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1, +1);
    wave_sync();
    const int base = chunk_base(c, +1), len = chunk_len(c, +1);
    double wv[kSbChunk];
#pragma unroll
    for (int q = 0; q < kSbChunk; ++q) wv[q] = (q < len) ? P.xs[(size_t)9 * (base + q) + lane] : 0.0;
    for (int q = 0; q < len; ++q) {
      const double* L = sblk[c & 1][q];
      double w = wv[0];
#pragma unroll
      for (int k = 1; k < kSbChunk; ++k) w = (q == k) ? wv[k] : w;
#pragma unroll
      for (int k = 0; k < 9; ++k) w -= L[9 * lane + k] * rdlane64(prev, k);  // Lsub of a chain head is zero
      double u = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) u += L[84 + 9 * lane + k] * rdlane64(w, k);  // Ldinv is lower triangular: entries k > lane are zero
      if (act) P.xs[(size_t)9 * (base + q) + lane] = u;
      prev = u;
    }
  }
  // ---- backward: x_pos = Linv^T (u_pos - Lsub_{pos+1}^T x_{pos+1}); needs Lsub of the position ABOVE, so a chunk
  //      keeps the previous chunk's buffer alive for its first step (double buffering does that for free)
  prev = 0.0;
  __syncthreads();  // (full barrier once: the backward sweep re-reads xs entries the forward sweep stored)
  gload(0, -1);
  for (int c = 0; c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1, -1);
    wave_sync();
    const int base = chunk_base(c, -1), len = chunk_len(c, -1);
    double uv[kSbChunk];
#pragma unroll
    for (int q = 0; q < kSbChunk; ++q) uv[q] = (q < len) ? P.xs[(size_t)9 * (base + q) + lane] : 0.0;
    for (int q = len - 1; q >= 0; --q) {
      const int pos = base + q;
      // Lsub_{pos+1}: next slot of this chunk, or slot 0 of the previous (higher) chunk, or nothing at the chain end
      const double* Lup = (q + 1 < len) ? sblk[c & 1][q + 1] : (c > 0 ? sblk[(c - 1) & 1][0] : nullptr);
      const double* L = sblk[c & 1][q];
      double w = uv[0];
#pragma unroll
      for (int k = 1; k < kSbChunk; ++k) w = (q == k) ? uv[k] : w;
      if (Lup != nullptr && pos + 1 < p1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) w -= Lup[9 * k + lane] * rdlane64(prev, k);
      }
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) x += L[84 + 9 * k + lane] * rdlane64(w, k);  // column of a lower-triangular matrix: entries k < lane are zero
      if (act) P.xs[(size_t)9 * pos + lane] = x;
      prev = x;
    }
  }
}

// right behind the chain factorisation: sweep matrices, f = Ldinv b_s, then z by the forward sweep (xs still holds b_s)
static void launch_sb_after_factor(const DevProblem& P, hipStream_t st) {
  hipLaunchKernelGGL(k_sb_sweep_mat, dim3((81 * P.K + 255) / 256), dim3(256), 0, st, P);
  hipLaunchKernelGGL(k_sb_sweep_vec<1>, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, (const double*)P.xs);
  hipLaunchKernelGGL(k_sb_sweep<1>, dim3(P.nchains), dim3(64), 0, st, P);
  // x0 = A^-1 b_s = L_A^-T z by the backward sweep (into xs: b_s is not needed any more), for the reduced pose right-hand
  // side b_p - B^T x0 (k_pose_rhs_sweep) — the Y^T z product it replaces read all of Y (430 MB, 0.25 ms on the critical path)
  hipLaunchKernelGGL(k_sb_sweep_vec<-1>, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, (const double*)P.zs);
  hipLaunchKernelGGL(k_sb_sweep<-1>, dim3(P.nchains), dim3(64), 0, st, P);
}
void launch_sb_chain_factor_early(const DevProblem& P, hipStream_t st, CholAux& ax) {
  if (!P.vi) return;
  ax.init();
  hipLaunchKernelGGL(k_gather_rhs, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, 1);
  (void)hipEventRecord(ax.ev_sb, st);
  (void)hipStreamWaitEvent(ax.aux, ax.ev_sb, 0);
  hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, ax.aux, P);
  hipLaunchKernelGGL(k_sb_propagator, dim3((81 * P.K + 255) / 256), dim3(256), 0, ax.aux, P);  // k_sb_chain_cols' tail marches with it
  (void)hipEventRecord(ax.ev_cf, ax.aux);
  // z = L_A^-1 b_s, x0 = A^-1 b_s (linear recurrences, k_sb_sweep) and the Gramians need the factor only: the sweeps follow
  // on the auxiliary stream, the Gramians run beside them on a third one, both underneath k_sb_chain_cols
  launch_sb_after_factor(P, ax.aux);
  (void)hipEventRecord(ax.ev_z, ax.aux);
  (void)hipStreamWaitEvent(ax.mid, ax.ev_cf, 0);
  hipLaunchKernelGGL(k_sb_gram, dim3(P.nchains), dim3(128), 0, ax.mid, P);
  (void)hipEventRecord(ax.ev_g, ax.mid);
  ax.cf_pending = true;
}

void launch_structured_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax, PgoPlan* pgo, NdDev* nd) {
  if (nd != nullptr && P.nd) { launch_nd_solve(P, *nd, dst, st, ax); return; }  // multifrontal form (k_front.hip): the whole system at once
  const int cnt = P.npad > 9 * P.K ? P.npad : 9 * P.K;
  const bool early = P.vi && ax.cf_pending;  // chain factor (and z in xs) already under way on the auxiliary stream
  hipLaunchKernelGGL(k_gather_rhs, dim3((cnt + 255) / 256), dim3(256), 0, st, P, early ? 0 : 2);
  if (P.vi) {
    if (early) { (void)hipStreamWaitEvent(st, ax.ev_cf, 0); ax.cf_pending = false; }
    else {
      hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, st, P);
      hipLaunchKernelGGL(k_sb_propagator, dim3((81 * P.K + 255) / 256), dim3(256), 0, st, P);
    }
    hipLaunchKernelGGL(k_sb_chain_cols, dim3(P.cc_n), dim3(256), 0, st, P);
    if (early) (void)hipStreamWaitEvent(st, ax.ev_g, 0);  // Gramians (third stream)
    else {
      launch_sb_after_factor(P, st);
      hipLaunchKernelGGL(k_sb_gram, dim3(P.nchains), dim3(128), 0, st, P);
    }
    hipLaunchKernelGGL(k_yty_semisep, dim3(P.K, (6 * P.max_chain_len + 255) / 256), dim3(256), 0, st, P);
    if (early) (void)hipStreamWaitEvent(st, ax.ev_z, 0);  // z, x0 (chain sweeps, auxiliary stream)
    static const bool rhs_from_y = [] { const char* e = getenv("COVGPU_POSE_RHS_Y"); return e && e[0] == '1'; }();  // 1: round-2a product with Y
    if (rhs_from_y) hipLaunchKernelGGL(k_pose_rhs, dim3((8 * 6 * P.K + 255) / 256), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(k_pose_rhs_sweep, dim3((6 * P.K + 255) / 256), dim3(256), 0, st, P);
  }
  if (pgo != nullptr) launch_pgo_block_solve(P, *pgo, st, ax);  // pose graph: block-arrow elimination (k_pgo.hip)
  else if (P.arrow) launch_arrow_solve(P, st, ax);                // GBA on a fused multi-agent map: block-arrow elimination (k_arrow.hip)
  else dense_cholesky_solve_raw(P.Sred, P.bp, P.Linv, P.flag, P.npad, st, ax);
  if (P.vi) {
    static const bool fwd_sweep = [] { const char* e = getenv("COVGPU_SB_FWD_SWEEP"); return e && e[0] == '1'; }();  // 1: round-1 sequential forward sweep
    if (fwd_sweep) {
      hipLaunchKernelGGL(k_sb_rhs, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P);
      hipLaunchKernelGGL(k_sb_backsolve, dim3(P.nchains), dim3(64), 0, st, P, 1);
    } else {
      hipLaunchKernelGGL(k_sb_fwd_matvec, dim3((9 * P.K + 3) / 4), dim3(256), 0, st, P);
      static const bool two_products = [] { const char* e = getenv("COVGPU_SB_BACK"); return e && e[0] == '0'; }();  // 0: round-2a backward sweep
      if (two_products) hipLaunchKernelGGL(k_sb_backsolve, dim3(P.nchains), dim3(64), 0, st, P, 0);
      else {
        hipLaunchKernelGGL(k_sb_sweep_vec<-1>, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, (const double*)P.xs);
        hipLaunchKernelGGL(k_sb_sweep<-1>, dim3(P.nchains), dim3(64), 0, st, P);
      }
    }
  }
  hipLaunchKernelGGL(k_scatter_solution, dim3((P.n + 255) / 256), dim3(256), 0, st, P, dst);
}

}  // namespace covgpu
