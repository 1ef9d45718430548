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
//                     Lsub_{k+1} = Ae_{k+1} L_kk^-T, z = L_A^-1 b_s (operands staged through LDS a chunk ahead)
//   sb_chain_cols     Y = L_A^-1 B, one thread per pose dimension marching down its chain (K-major, coalesced)
//   pose_rhs          b'_p = b_p - Y^T z
//   (k_chol.hip)      C -= Y^T Y restricted to each tile pair's common chain segment; dense Cholesky of C'
//   sb_backsolve      x_s = A^-1 (b_s - B x_p) with the stored bidiagonal factor, one wave per chain
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
constexpr int kCfChunk = 8;                     // positions staged per LDS buffer
constexpr int kCfBlk = 81 + 81 + 9;             // Ad_pos | Ae_{pos+1} | b_s,pos
constexpr int kCfPerLane = (kCfChunk * kCfBlk + 63) / 64;
__global__ __launch_bounds__(64) void k_sb_chain_factor(DevProblem P) {
  __shared__ double sM[81], sX[81], sSub[81];
  __shared__ double sz[9], sv[9];
  __shared__ double sIn[2][kCfChunk][kCfBlk];   // operands, staged kCfChunk positions ahead by all 64 lanes
  const int lane = threadIdx.x;
  const int p0 = P.chain_ptr[blockIdx.x], p1 = P.chain_ptr[blockIdx.x + 1];
  const int e0 = lane, e1 = lane + 64;  // the two matrix entries this lane owns (e1 valid for lanes < 17)
  const bool has1 = e1 < 81;
  for (int e = lane; e < 81; e += 64) sSub[e] = 0.0;
  if (lane < 9) sz[lane] = 0.0;
  // Operand loads one step ahead ran at HBM latency under the landmark pass's traffic (5.4 us per step); the next
  // chunk's loads are now in flight during a whole chunk of steps.
  const int nchunk = (p1 - p0 + kCfChunk - 1) / kCfChunk;
  double stage[kCfPerLane];
  auto gload = [&](int c) {
    const int base = p0 + c * kCfChunk;
#pragma unroll
    for (int i = 0; i < kCfPerLane; ++i) {
      const int idx = lane + 64 * i, q = idx / kCfBlk, e = idx - kCfBlk * q, pos = base + q;
      double v = 0.0;
      if (q < kCfChunk && pos < p1) {
        if (e < 81) v = P.Ad[(size_t)81 * pos + e];
        else if (e < 162) v = (pos + 1 < p1) ? P.Ae[(size_t)81 * (pos + 1) + e - 81] : 0.0;
        else v = P.xs[(size_t)9 * pos + e - 162];
      }
      stage[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kCfPerLane; ++i) {
      const int idx = lane + 64 * i, q = idx / kCfBlk, e = idx - kCfBlk * q;
      if (q < kCfChunk) sIn[buf][q][e] = stage[i];
    }
  };
  gload(0);
  for (int c = 0; c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1);
    __syncthreads();
    const int base = p0 + c * kCfChunk, len = min(kCfChunk, p1 - base);
    for (int q = 0; q < len; ++q) {
    const int pos = base + q;
    const double* in = sIn[c & 1][q];
    const double* sAe = in + 81;
    const double ad0 = in[e0], ad1 = has1 ? in[e1] : 0.0;
    const double bs = (lane < 9) ? in[162 + lane] : 0.0;
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
    // lower Cholesky, row r = lane (lanes >= 9 carry zeros); then X = L^-1 column by column, still in registers:
    // lane c solves L x = e_c with the entries of L broadcast by v_readlane and the reciprocal pivots the Cholesky
    // already has (no divisions, no LDS round trip between the two phases)
    {
      const int r = lane < 9 ? lane : 8;
      double x[9], invd[9];
#pragma unroll
      for (int c2 = 0; c2 < 9; ++c2) x[c2] = (lane < 9 && c2 <= r) ? sM[9 * r + c2] : 0.0;
      bool bad = false;
#pragma unroll
      for (int c2 = 0; c2 < 9; ++c2) {
        double d = rdlane64(x[c2], c2);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double inv = rsqrt(d);
        invd[c2] = inv;  // wave-uniform
        x[c2] = (lane == c2) ? d * inv : x[c2] * inv;
#pragma unroll
        for (int cc = c2 + 1; cc < 9; ++cc) x[cc] -= x[c2] * rdlane64(x[c2], cc);  // garbage above the diagonal is never read
      }
      if (bad && lane == 0) atomicOr(P.flag, 1);
      double xi[9];  // column `lane` of X
#pragma unroll
      for (int rr = 0; rr < 9; ++rr) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) sum += rdlane64(x[k], rr) * xi[k];  // L[rr][k]; xi[k] is 0 for k < lane
        xi[rr] = (rr == lane) ? invd[rr] : (rr > lane ? -sum * invd[rr] : 0.0);
      }
      __syncthreads();  // everyone has read sM
      if (lane < 9) {
#pragma unroll
        for (int c2 = 0; c2 < 9; ++c2) sM[9 * lane + c2] = (c2 <= lane) ? x[c2] : 0.0;
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) sX[9 * rr + lane] = xi[rr];
        double v = bs;
#pragma unroll
        for (int k = 0; k < 9; ++k) v -= sSub[9 * lane + k] * sz[k];
        sv[lane] = v;
      }
    }
    __syncthreads();
    // publish the factor blocks, z, and the next sub-diagonal block L_{pos+1,pos} = Ae_{pos+1} L_kk^-T
    P.Ldinv[(size_t)81 * pos + e0] = sX[e0]; P.Lsub[(size_t)81 * pos + e0] = sSub[e0];  // (L_kk itself is needed only through its inverse)
    if (has1) { P.Ldinv[(size_t)81 * pos + e1] = sX[e1]; P.Lsub[(size_t)81 * pos + e1] = sSub[e1]; }
    double znew = 0.0;
    if (lane < 9) {
      for (int k = 0; k <= lane; ++k) znew += sX[9 * lane + k] * sv[k];
      P.zs[(size_t)9 * pos + lane] = znew;
    }
    double nx0 = 0.0, nx1 = 0.0;
    if (pos + 1 < p1) {
      {
        const int a = e0 / 9, b = e0 - 9 * a;
#pragma unroll
        for (int c2 = 0; c2 < 9; ++c2) nx0 += sAe[9 * a + c2] * sX[9 * b + c2];
      }
      if (has1) {
        const int a = e1 / 9, b = e1 - 9 * a;
#pragma unroll
        for (int c2 = 0; c2 < 9; ++c2) nx1 += sAe[9 * a + c2] * sX[9 * b + c2];
      }
    }
    __syncthreads();  // sSub, sz, sv are free: every lane has finished reading them
    sSub[e0] = nx0;
    if (has1) sSub[e1] = nx1;
    if (lane < 9) sz[lane] = znew;
    __syncthreads();
    }
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
  ColOps o0 = fetch(pfirst), o1 = fetch(pfirst + 1), o2 = fetch(pfirst + 2), o3 = fetch(pfirst + 3);
  v4f64s y = {0.0, 0.0, 0.0, 0.0};
  for (int pos = pfirst; pos < p1; ++pos) {
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
    if (live && pos >= pstart) {
      double* Yp = Yc + (size_t)(9 * (pos - p0) + kq) * ld + (col - 6 * p0);
      Yp[0] = y[0];
      Yp[4 * ld] = y[1];
      if (kq == 0) Yp[8 * ld] = y[2];
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
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) t += sG[9 * a + k] * M[9 * k + b];   // T = (I + G_{q+1}) M
    if (act) sT[tid] = t;
    __syncthreads();
    double g = eye;
#pragma unroll
    for (int k = 0; k < 9; ++k) g += M[9 * k + a] * sT[9 * k + b];   // I + M^T T
    if (act) { sG[tid] = g; P.GI[(size_t)81 * q + tid] = g; }
    // (the next iteration writes the OTHER sM buffer; its first barrier orders this sG write before the reads)
  }
}

// One workgroup per pose j: C[rows 6j..6j+5, cols of poses <= j in the chain] -= W_j^T Y[rows 9(j-1) .. 9(j+2), cols]
// with W_j = [y_j(j-1); y_j(j); (I + G_{j+1}) y_j(j+1)] (27 x 6, built in LDS first).
__global__ __launch_bounds__(256) void k_yty_semisep(DevProblem P) {
  __shared__ double sW[27][6];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int p1 = P.pos_chain_end[j], p0 = P.pos_chain_begin[j], ch = P.pos_chain[j];
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
  for (int c = 6 * p0 + tid; c < 6 * (j + 1); c += 256) {
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
    __syncthreads();
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
  __syncthreads();
  gload(0, -1);
  for (int c = 0; c < nchunk; ++c) {
    lstore(c & 1);
    if (c + 1 < nchunk) gload(c + 1, -1);
    __syncthreads();
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

void launch_sb_chain_factor_early(const DevProblem& P, hipStream_t st, CholAux& ax) {
  if (!P.vi) return;
  ax.init();
  hipLaunchKernelGGL(k_gather_rhs, dim3((9 * P.K + 255) / 256), dim3(256), 0, st, P, 1);
  (void)hipEventRecord(ax.ev_sb, st);
  (void)hipStreamWaitEvent(ax.aux, ax.ev_sb, 0);
  hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, ax.aux, P);
  (void)hipEventRecord(ax.ev_cf, ax.aux);
  // the Gramians need the factor only: they follow on the auxiliary stream, underneath k_sb_chain_cols
  hipLaunchKernelGGL(k_sb_propagator, dim3((81 * P.K + 255) / 256), dim3(256), 0, ax.aux, P);
  hipLaunchKernelGGL(k_sb_gram, dim3(P.nchains), dim3(128), 0, ax.aux, P);
  (void)hipEventRecord(ax.ev_g, ax.aux);
  ax.cf_pending = true;
}

void launch_structured_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax, PgoPlan* pgo) {
  const int cnt = P.npad > 9 * P.K ? P.npad : 9 * P.K;
  const bool early = P.vi && ax.cf_pending;  // chain factor (and z in xs) already under way on the auxiliary stream
  hipLaunchKernelGGL(k_gather_rhs, dim3((cnt + 255) / 256), dim3(256), 0, st, P, early ? 0 : 2);
  if (P.vi) {
    if (early) { (void)hipStreamWaitEvent(st, ax.ev_cf, 0); ax.cf_pending = false; }
    else hipLaunchKernelGGL(k_sb_chain_factor, dim3(P.nchains), dim3(64), 0, st, P);
    hipLaunchKernelGGL(k_sb_chain_cols, dim3(P.cc_n), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_pose_rhs, dim3((8 * 6 * P.K + 255) / 256), dim3(256), 0, st, P);
    if (early) (void)hipStreamWaitEvent(st, ax.ev_g, 0);
    else {
      hipLaunchKernelGGL(k_sb_propagator, dim3((81 * P.K + 255) / 256), dim3(256), 0, st, P);
      hipLaunchKernelGGL(k_sb_gram, dim3(P.nchains), dim3(128), 0, st, P);
    }
    hipLaunchKernelGGL(k_yty_semisep, dim3(P.K), dim3(256), 0, st, P);
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
      hipLaunchKernelGGL(k_sb_backsolve, dim3(P.nchains), dim3(64), 0, st, P, 0);
    }
  }
  hipLaunchKernelGGL(k_scatter_solution, dim3((P.n + 255) / 256), dim3(256), 0, st, P, dst);
}

}  // namespace covgpu
