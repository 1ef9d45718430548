// k_chol.hip — dense FP64 Cholesky solve of the reduced camera system on the gfx950 matrix cores.
//
// Replaces the linear-algebra half of ceres::Solve(SPARSE_SCHUR) (optimization_be.cpp:560-567, 1024-1031):
// the Cholesky factorisation of the reduced camera system (CHOLMOD in the reference) and the two triangular
// solves. North-star: "MFMA used only for the dense reduced-camera-system solve".
//
// Algorithm: right-looking blocked Cholesky on the lower triangle of a row-major padded matrix.
//   * tiles are 128 x 128; big panels are 256 wide (two tile columns). A rank-256 trailing update reads and
//     writes every C tile once per 8.4 MFLOP (16 flop/B on the C stream) — a rank-128 update (8 flop/B)
//     would be HBM-bound below ~60 % of the FP64 MFMA peak.
//   * panel factorisation (default, k_panel.hip): ONE workgroup factors the whole 256x256 diagonal block (register-resident
//     16x16 tiles, software-pipelined around a single wave that carries the serial chain), the rows below follow by block
//     forward substitution against the sixteen 16x16 block inverses it leaves — three dependent launches per panel.
//     COVGPU_PANEL=0 selects the earlier chain kept in this file: potrf_inv (128x128 block + explicit inverse) -> TRSM as an
//     MFMA GEMM against L11^-1 -> rank-128 update of the panel's second tile column -> potrf_inv -> TRSM (six launches).
//   * trailing update k_gemm_abt<SYRK_TRI>: C -= A_i A_j^T, one 128x128 tile per workgroup, 4 waves x (4x4)
//     v_mfma_f64_16x16x4_f64 tiles, accumulators initialised FROM the C tile (so the epilogue is store-only),
//     K staged through LDS in chunks of 16 with register prefetch. Workgroup ids are decoded XCD-aware: the 64
//     workgroups resident on one XCD at a time form one 8x8 supertile, so its 16 panel tiles stay in that
//     XCD's 4 MiB L2 instead of being re-fetched by every tile (block b runs on XCD b % 8).
//   * look-ahead on four HIP streams (dense_cholesky_solve_raw): the serial chain potrf -> X(t0+1,t0) -> D1 update
//     -> potrf -> next diagonal update alone on the main stream, the next panel's tile rows, all rows below, and the
//     bulk of the trailing update on three more; the chain's few-tile GEMMs run as quarter tiles.
//   * entry points beyond the plain solve: partial factorisation (tstop: the trailing block keeps its Schur
//     complement), a batch dimension (independent matrices of one shape in the same launches) and a backward solve
//     with given trailing unknowns — the block-arrow pose-graph solve (k_pgo.hip) is built from these.
// Forward substitution rides along with the factorisation (potrf forms y_p, every TRSM takes its tile's product
// with y_p out of the right-hand side); the backward solve reuses the stored L_pp^-1 blocks, one launch per panel.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
using namespace covdev;

typedef double v4f64 __attribute__((ext_vector_type(4)));

// phase timers for tools/potrf_probe.hip (never defined in the product build)
#ifdef COVGPU_PROBE
__device__ long long g_probe[8];
#define PROBE(i) do { if (threadIdx.x == 0) g_probe[i] = wall_clock64(); } while (0)
#define PROBE_ACC(i, t0) do { if (threadIdx.x == 0) g_probe[i] += wall_clock64() - (t0); } while (0)
#define PROBE_T0() wall_clock64()
#else
#define PROBE(i) do {} while (0)
#define PROBE_ACC(i, t0) do {} while (0)
#define PROBE_T0() 0
#endif

constexpr int KC = 16;        // K chunk staged through LDS (full-tile kernels)
constexpr int KCQ = 32;       // quarter forms: they run at memory LATENCY (one dependent global->LDS step per chunk while the
                              // chip is loaded: ~3 us each), so fewer, larger chunks
constexpr int LDT = KC + 1;   // LDS pitch (doubles): odd pitch -> conflict-free fragment reads

enum { MODE_SYRK_TRI = 0, MODE_SYRK_RECT = 1, MODE_TRSM = 2 };

struct GemmArgs {
  double* M; size_t ld;
  int kcol0, KD;      // K range: columns kcol0 .. kcol0+KD of the panel (KD = 128 or 256)
  int ra0;            // A rows: ra0 + ti*128
  int rb0;            // B rows: rb0 + tj*128           (TRI: == ra0)
  int cc0;            // C tile at (ra0 + ti*128, cc0 + tj*128)   (TRI: == ra0; TRSM: == kcol0, tj = 0)
  int nt;             // tile rows (TRI: triangle order)
  const double* Linv; // TRSM: B = Linv (128x128, pitch 128)
  // TRSM with a right-hand side riding along (forward substitution fused into the factorisation):
  // rhs[rows of this workgroup] -= X[rows, :] yvec[kcol0 .. kcol0+128), X = the freshly computed L tile
  double* rhs; const double* yvec;
  // batched form (block-arrow pose-graph solve): independent matrices of identical shape, `batch` strides apart
  size_t bsM, bsL, bsR;  // elements between consecutive matrices / Linv sets / right-hand sides (0: not batched)
  // arrow buffers of unequal blocks are padded to one shape: live[2*batch] = real interior tiles, live[2*batch+1] = real
  // own-border tiles (which start at tile tI). A tile row outside both ranges is identity / zero padding: every product
  // with it is zero, so workgroups that would only touch padding exit at once (about half of the batched flops on the
  // 5-agent map, whose agents hold 267 .. 498 interior keyframes).
  const int* live; int tI;
  // SYRK_TRI on arrow buffers: explicit list of the live tiles, entry = batch << 20 | ti << 10 | tj (-1: no work)
  const int* tri;
  // fronts of unequal order in one batch (k_front.hip): btab[2 batch] = element offset of the matrix, btab[2 batch + 1] = its
  // leading dimension; nullptr = the uniform bsM / ld above
  const long long* btab;
};
__device__ __forceinline__ bool tile_live(const GemmArgs& g, int batch, int t) {
  if (g.live == nullptr) return true;
  const int nI = g.live[2 * batch], nO = g.live[2 * batch + 1];
  return t < nI || (t >= g.tI && t - g.tI < nO);
}

// C[i][j] (op)= sum_k A[i][k] B[j][k] on one 128x128 tile (TSA = TSB = 128), or on a quarter of it selected by
// blockIdx.z — a 64x64 quadrant (RECT) or a 32x128 row slab (TRSM: the update is in place, X overwrites A, so a
// workgroup must own whole rows). The quarter forms are for the few-tile launches on the serial panel chain, where
// the latency of ONE workgroup's tile is the whole cost — 4x more workgroups, each 4x shorter.
//
// All global addresses are a wave-uniform base (SGPRs) plus a 32-bit per-lane byte offset instead of 64-bit per-lane
// pointers (8 fewer VGPR pairs to carry and to bump per chunk).
// Measured, not adopted (tools/dispatch_probe.hip): a small kernel launched while the chip is full
// of bulk workgroups starts AT ONCE if its waves fit beside the resident ones (2 x bulk + small <= 512 VGPRs per
// SIMD lane: 5.8 us per dependent launch) and otherwise queues for a bulk workgroup to retire (34.8 us). Capping
// these kernels with amdgpu_num_vgpr (the attribute counts HALF registers on gfx90a+: 104 -> 208 for the full
// tiles, 48 -> 96 for the quarter forms; spills stay outside the K loop) makes them fit — and changes nothing
// end to end (27.85 vs 27.65 ms, bulk 41.0 vs 42.5 TFLOP/s): under load the quarter kernels are bound by their
// 4-16 dependent global -> LDS steps at loaded memory latency, not by the dispatch.
template <int MODE, int TSA, int TSB, int KC>
COV_DEV void gemm_abt_body(const GemmArgs& g) {
  constexpr int LDT = KC + 1;
  static_assert((TSA == kTile && TSB == kTile) || (MODE == MODE_SYRK_RECT && TSA == 64 && TSB == 64) ||
                (MODE == MODE_TRSM && TSA == 32 && TSB == kTile), "quarter forms: RECT 64x64, TRSM 32x128");
  constexpr int WGR = (TSA == 32) ? 1 : 2, WGC = 4 / WGR;  // wave grid
  constexpr int WTR = TSA / WGR, WTC = TSB / WGC;           // wave tile
  constexpr int NMR = WTR / 16, NMC = WTC / 16;             // MFMA tiles per wave
  constexpr int ZQ = (TSA == kTile && TSB == kTile) ? 1 : 4;  // quarter index and batch index share blockIdx.z
  if (ZQ == 4) __builtin_amdgcn_s_setprio(3);  // quarter forms run on the serial chain only, beside bulk waves: issue priority over them
  int tri_entry = 0;
  if (MODE == MODE_SYRK_TRI && g.tri != nullptr) { tri_entry = g.tri[blockIdx.x]; if (tri_entry < 0) return; }
  const int zq = (int)blockIdx.z % ZQ,
            batch = (MODE == MODE_SYRK_TRI) ? (g.tri != nullptr ? (tri_entry >> 20) : (int)blockIdx.y) : (int)blockIdx.z / ZQ;
  const int qr = (TSA == kTile) ? 0 : (TSB == kTile ? zq : (zq >> 1));
  const int qc = (TSB == kTile) ? 0 : (zq & 1);
  double* const Mb = g.M + (g.btab != nullptr ? (size_t)g.btab[2 * batch] : (size_t)batch * g.bsM);
  int ti, tj;
  if (MODE == MODE_SYRK_TRI && g.tri != nullptr) {
    ti = (tri_entry >> 10) & 1023; tj = tri_entry & 1023;
  } else if (MODE == MODE_SYRK_TRI) {
    // XCD-aware decode: block b -> XCD (b & 7) (observed dispatch order; placement affects speed only).
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int s = (q >> 6) * 8 + xcd, inner = q & 63;
    int si = (int)((sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
    while ((si + 1) * (si + 2) / 2 <= s) ++si;
    while (si * (si + 1) / 2 > s) --si;
    const int sj = s - si * (si + 1) / 2;
    ti = si * 8 + (inner >> 3); tj = sj * 8 + (inner & 7);
    if (ti >= g.nt || tj > ti) return;
  } else if (MODE == MODE_SYRK_RECT) {
    ti = blockIdx.y; tj = blockIdx.x;
    if (g.ra0 + ti * kTile < g.cc0 + tj * kTile) return;  // strictly above the diagonal
  } else {
    ti = blockIdx.x; tj = 0;
  }
  if (g.live != nullptr) {  // wave-uniform early exit on padding (see GemmArgs::live)
    if (!tile_live(g, batch, g.kcol0 / kTile)) return;                       // the panel itself is padding: A = B = 0
    if (!tile_live(g, batch, g.ra0 / kTile + ti)) return;                    // A rows are padding
    if (MODE != MODE_TRSM && !tile_live(g, batch, g.rb0 / kTile + tj)) return;  // B rows are padding
  }
  const int kbeg = 0, kend = g.KD;
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [TSA][KC+1] + [TSB][KC+1] doubles
  double (*sA)[LDT] = reinterpret_cast<double (*)[LDT]>(smem);
  double (*sB)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + TSA * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WGC, wc = wave % WGC;
  const size_t ld = g.btab != nullptr ? (size_t)g.btab[2 * batch + 1] : g.ld;
  // wave-uniform bases (the tile of a workgroup spans < 2^32 bytes: per-lane offsets are 32-bit byte offsets)
  const char* Ag = reinterpret_cast<const char*>(Mb + (size_t)(g.ra0 + ti * kTile + qr * TSA) * ld + g.kcol0);
  const char* Bg = (MODE == MODE_TRSM) ? reinterpret_cast<const char*>(g.Linv + (size_t)batch * g.bsL)
                                       : reinterpret_cast<const char*>(Mb + (size_t)(g.rb0 + tj * kTile + qc * TSB) * ld + g.kcol0);
  const unsigned ldab = (unsigned)(ld * sizeof(double)), ldbb = (MODE == MODE_TRSM) ? (unsigned)(kTile * sizeof(double)) : ldab;
  char* Cg = reinterpret_cast<char*>(Mb + (size_t)(g.ra0 + ti * kTile + qr * TSA) * ld + (size_t)(g.cc0 + tj * kTile + qc * TSB));
  // staging map: KC/2 lanes cover one KC-double row segment (contiguous), 512/KC rows per pass
  constexpr int LPR = KC / 2, RPP = 256 / LPR, NPA = TSA / RPP, NPB = TSB / RPP;
  const int c2 = (tid % LPR) * 2, rbase = tid / LPR;
  const unsigned offa = (unsigned)rbase * ldab + (unsigned)c2 * 8u, offb = (unsigned)rbase * ldbb + (unsigned)c2 * 8u;
  double2 pa[NPA], pb[NPB];
  auto gload = [&](int kc) {
    const char* Ak = Ag + (size_t)kc * sizeof(double);  // uniform
    const char* Bk = Bg + (size_t)kc * sizeof(double);
#pragma unroll
    for (int it = 0; it < NPA; ++it) pa[it] = *reinterpret_cast<const double2*>(Ak + (size_t)(RPP * it) * ldab + offa);
#pragma unroll
    for (int it = 0; it < NPB; ++it) pb[it] = *reinterpret_cast<const double2*>(Bk + (size_t)(RPP * it) * ldbb + offb);
  };
  gload(kbeg);
  const int fr = lane & 15, fk = lane >> 4;
  // f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  const unsigned offc = (unsigned)(wr * WTR + fk) * ldab + (unsigned)(wc * WTC + fr) * 8u;  // per lane; the rest is uniform
  v4f64 acc[NMR][NMC];
#pragma unroll
  for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const unsigned rowoff = offc + (unsigned)(tm * 16 + 4 * rg) * ldab;  // one VGPR per row; tn goes into the immediate
#pragma unroll
      for (int tn = 0; tn < NMC; ++tn) {
        if (MODE == MODE_TRSM) acc[tm][tn][rg] = 0.0;
        else acc[tm][tn][rg] = *reinterpret_cast<const double*>(Cg + rowoff + tn * 128);
      }
    }
  const double sgn = (MODE == MODE_TRSM) ? 1.0 : -1.0;  // SYRK: acc = C - A B^T through a negated A fragment
  for (int kc = kbeg; kc < kend; kc += KC) {
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int it = 0; it < NPA; ++it) { sA[rbase + RPP * it][c2] = sgn * pa[it].x; sA[rbase + RPP * it][c2 + 1] = sgn * pa[it].y; }
#pragma unroll
    for (int it = 0; it < NPB; ++it) { sB[rbase + RPP * it][c2] = pb[it].x; sB[rbase + RPP * it][c2 + 1] = pb[it].y; }
    __syncthreads();
    if (kc + KC < kend) gload(kc + KC);  // prefetch the next chunk while the matrix cores work
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      double a[NMR], b[NMC];
#pragma unroll
      for (int t = 0; t < NMR; ++t) a[t] = sA[wr * WTR + t * 16 + fr][kk + fk];
#pragma unroll
      for (int t = 0; t < NMC; ++t) b[t] = sB[wc * WTC + t * 16 + fr][kk + fk];
#pragma unroll
      for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
        for (int tn = 0; tn < NMC; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
  }
  // (TRSM in place: every A element of this workgroup's rows was staged before the last barrier above)
#pragma unroll
  for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const unsigned rowoff = offc + (unsigned)(tm * 16 + 4 * rg) * ldab;
#pragma unroll
      for (int tn = 0; tn < NMC; ++tn) *reinterpret_cast<double*>(Cg + rowoff + tn * 128) = acc[tm][tn][rg];
    }
  if (MODE == MODE_TRSM && g.rhs != nullptr) {
    // forward substitution riding along: rhs[row] -= sum_c X[row][c] y[c]. Lane partial over its columns, fixed
    // butterfly over the 16 lanes that share a row, fixed-order sum over the wave columns: deterministic.
    __syncthreads();                     // the staging buffers are free
    double* sy = smem;                   // [128] y of this panel
    double* sp = smem + kTile;           // [WGC][TSA] partial row sums
    if (tid < kTile) sy[tid] = g.yvec[(size_t)batch * g.bsR + g.kcol0 + tid];
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        double pr = 0.0;
#pragma unroll
        for (int tn = 0; tn < NMC; ++tn) pr += acc[tm][tn][rg] * sy[wc * WTC + tn * 16 + fr];
        pr += __shfl_xor(pr, 1, 64); pr += __shfl_xor(pr, 2, 64); pr += __shfl_xor(pr, 4, 64); pr += __shfl_xor(pr, 8, 64);
        if (fr == 0) sp[wc * TSA + wr * WTR + tm * 16 + fk + 4 * rg] = pr;
      }
    __syncthreads();
    if (tid < TSA) {
      double t = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < WGC; ++w2) t += sp[w2 * TSA + tid];
      g.rhs[(size_t)batch * g.bsR + g.ra0 + ti * kTile + qr * TSA + tid] -= t;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_gemm_abt(GemmArgs g) { gemm_abt_body<MODE, kTile, kTile, KC>(g); }
template <int MODE, int TSA, int TSB>
__global__ __launch_bounds__(256, 2) void k_gemm_abt_q(GemmArgs g) { gemm_abt_body<MODE, TSA, TSB, KCQ>(g); }

COV_DEV double rdlane64c(double v, int srclane) {  // broadcast from a wave-uniform lane through SGPRs
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), srclane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Shared tail of the potrf kernels. On entry `s` ([128][129] in LDS) holds, in its 16x16 diagonal blocks, the INVERSES of
// the factor's diagonal blocks and, below them, the factor L itself; on exit Linv_out = L^-1 and (optionally) y = L^-1 rhs.
// one level of the recursive-doubling inverse, NQ tiles per wave advanced in lockstep: NQ independent accumulator chains
// keep the matrix pipe busy where a single dependent chain of v_mfma_f64 left it idle most of the time (tools/potrf_probe:
// 20 us for the three levels with one chain per wave)
template <int NQ>
COV_DEV void inv_level(double* s, int h, int wave, int fr, int fk) {
  constexpr int PT = kTile + 1;
  const int hb = h >> 4, tpp = hb * hb;
  int base[NQ], tr[NQ], tc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int t = wave + 4 * q, pr = t / tpp, rem = t - pr * tpp;
    tr[q] = rem / hb; tc[q] = rem - tr[q] * hb; base[q] = 2 * pr * h;
  }
  __syncthreads();
  {  // T = C A^-1   (A^-1 lower: rows k >= c only); parked in the mirrored upper block
    v4f64 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < h; kk += 4) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (kk < tc[q] * 16) continue;  // wave-uniform
        const int k = kk + fk, c = tc[q] * 16 + fr;
        const double av = s[(base[q] + h + tr[q] * 16 + fr) * PT + base[q] + k];
        const double bv = (k >= c) ? s[(base[q] + k) * PT + base[q] + c] : 0.0;
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) s[(base[q] + tc[q] * 16 + fr) * PT + base[q] + h + tr[q] * 16 + fk + 4 * rg] = acc[q][rg];
  }
  __syncthreads();
  {  // X21 = -B^-1 T   (B^-1 lower: k <= r only); X overwrites the C block (no longer needed)
    v4f64 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < h; kk += 4) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (kk >= tr[q] * 16 + 16) continue;
        const int k = kk + fk, r = tr[q] * 16 + fr;
        const double av = (k <= r) ? s[(base[q] + h + r) * PT + base[q] + h + k] : 0.0;
        const double bv = s[(base[q] + tc[q] * 16 + fr) * PT + base[q] + h + k];
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) s[(base[q] + h + tr[q] * 16 + fk + 4 * rg) * PT + base[q] + tc[q] * 16 + fr] = -acc[q][rg];
  }
}

// Shared tail of the potrf kernels. On entry `s` ([128][129] in LDS) holds, in its 16x16 diagonal blocks, the INVERSES of
// the factor's diagonal blocks and, below them, the factor L itself; on exit Linv_out = L^-1 and (optionally) y = L^-1 rhs.
COV_DEV void potrf_tail(double* s, double* __restrict__ Linv_out, const double* __restrict__ rhs, double* __restrict__ yout, int k0) {
  constexpr int PT = kTile + 1;
  const int tid = threadIdx.x;
  // levels h = 16, 32, 64 on the matrix core: both products are small GEMMs (triangular operands masked to zero);
  // 4, 8, 16 output tiles per level = 1, 2, 4 per wave
  {
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    inv_level<1>(s, 16, wave, fr, fk);
    inv_level<2>(s, 32, wave, fr, fk);
    inv_level<4>(s, 64, wave, fr, fk);
  }
  __syncthreads();
  PROBE(4);
  for (int idx = tid; idx < kTile * kTile; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    Linv_out[idx] = (c <= r) ? s[r * PT + c] : 0.0;
  }
  // forward substitution riding along: y_p = L_pp^-1 b_p. Every earlier panel's TRSM has already taken its
  // L[rows p, q] y_q out of b_p (the diagonal tile itself depends on those TRSMs).
  if (rhs != nullptr && tid < kTile) {
    double acc = 0.0;
    for (int k = 0; k <= tid; ++k) acc += s[tid * PT + k] * rhs[k0 + k];
    yout[k0 + tid] = acc;
  }
}

// Factor the 128x128 diagonal block at (k0,k0) (lower Cholesky) and form its inverse.
// L (lower incl. diagonal) is written back into M; L^-1 (lower, zeros above) goes to Linv_out [128][128].
//
// Cholesky: right-looking, the matrix lives in REGISTERS — thread (ty,tx) of the 16x16 grid owns the 8x8 elements
// (r = ty + 16 i, c = tx + 16 k); only the pivot column travels through LDS each step. (Keeping the matrix in LDS
// and updating it in place serialises on LDS read-after-write: measured 260 us per block instead of ~40.)
__global__ __launch_bounds__(256) void k_potrf_inv(double* __restrict__ M, size_t ld, int k0, double* __restrict__ Linv_out, int* flag,
                                                    const double* __restrict__ rhs, double* __restrict__ yout, size_t bsM, size_t bsL, size_t bsR) {
  M += (size_t)blockIdx.x * bsM; Linv_out += (size_t)blockIdx.x * bsL;  // batched form: one workgroup per matrix
  if (rhs != nullptr) { rhs += (size_t)blockIdx.x * bsR; yout += (size_t)blockIdx.x * bsR; }
  extern __shared__ __attribute__((aligned(16))) double s[];  // [128][129]
  __shared__ __attribute__((aligned(16))) double colb[2][16][8];
  constexpr int PT = kTile + 1;
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  double* Mg = M + (size_t)k0 * ld + k0;
  double a[8][8];
  PROBE(0);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = ty + 16 * i, c = tx + 16 * k;
      a[i][k] = (c <= r) ? Mg[(size_t)r * ld + c] : 0.0;
    }
  PROBE(1);
  // (A variant that retires FOUR pivot columns per barrier pair — 4x4 diagonal block factored redundantly by every
  //  thread, panel rows by forward substitution on values broadcast inside each 16-lane row group — was measured
  //  SLOWER: 91 us against 61 us for this phase. One wave per SIMD is instruction-issue bound, and the four chained
  //  rsqrt plus 64 ds_bpermute per step cost more than the three barrier pairs they save.)
  // One barrier per pivot column: every thread keeps its own copy dd[k] of the diagonal entries of ITS columns
  // (updated with the same fused multiply-adds as the matrix entry itself, so the copies are bit-identical); the
  // owners of column j therefore know the pivot without a broadcast. The scaled column is double-buffered in LDS,
  // stored so that a thread's 8 row values (and its 8 column values) are contiguous: [r & 15][r >> 4].
  double dd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) dd[k] = Mg[(size_t)(tx + 16 * k) * ld + tx + 16 * k];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {  // unrolled: static register indices; the 16-step inner loop must stay rolled
#pragma unroll 1                    // (fully unrolled it is 22k instructions — far beyond the 64 KiB instruction cache)
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * jb + jj;
      double (*cb)[8] = colb[jj & 1];
      if (tx == jj) {  // owners of column j
        double d = dd[jb];
        if (!(d > 0.0)) { if (ty == 0) atomicOr(flag, 1); d = 1.0; }
        const double inv = rsqrt(d), sd = d * inv;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = ty + 16 * i;
          if (r == j) a[i][jb] = sd;
          else if (r > j) a[i][jb] *= inv;
          cb[ty][i] = (r > j) ? a[i][jb] : 0.0;
        }
      }
      __syncthreads();
      double cr[8], cc[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 u = *reinterpret_cast<const double2*>(&cb[ty][2 * q]);
        const double2 v = *reinterpret_cast<const double2*>(&cb[tx][2 * q]);
        cr[2 * q] = u.x; cr[2 * q + 1] = u.y; cc[2 * q] = v.x; cc[2 * q + 1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k)  // columns c > j only (the column is 0 for rows <= j); register tiles with k > i lie above the diagonal
          if (k <= i && (k > jb || (k == jb && tx > jj))) a[i][k] -= cr[i] * cc[k];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k >= jb) dd[k] -= cc[k] * cc[k];
    }
  }
  PROBE(2);
  // L -> LDS (lower) and back to HBM
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = ty + 16 * i, c = tx + 16 * k;
      s[r * PT + c] = (c <= r) ? a[i][k] : 0.0;
      if (c <= r) Mg[(size_t)r * ld + c] = a[i][k];
    }
  __syncthreads();
  PROBE(3);
  // ---- inverse by recursive doubling: 8x8 diagonal blocks in registers, then for h = 8,16,32,64 every pair
  //      [[A,0],[C,B]] -> [[A^-1,0],[-B^-1 C A^-1, B^-1]]. T = C A^-1 is parked in the (unused) mirrored upper
  //      block, X21 overwrites C. All dot products are independent: no serial LDS chain longer than h.
  {
    double Lb[8][8];
    const int c = tid & 127, cb = c & ~7, lc = c & 7;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k <= r; ++k) Lb[r][k] = s[(cb + r) * PT + cb + k];
    __syncthreads();
    if (tid < kTile) {
      double x[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) sum += Lb[r][k] * x[k];
        x[r] = (r == lc) ? 1.0 / Lb[r][r] : (r > lc ? -sum / Lb[r][r] : 0.0);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r >= lc) s[(cb + r) * PT + c] = x[r];
    }
  }
  {  // level h = 8 (8 pairs of 8x8 blocks): scalar dot products
    constexpr int h = 8;
    __syncthreads();
    for (int idx = tid; idx < 64 * h; idx += 256) {  // T = C A^-1
      const int pr = idx >> 6, rem = idx & 63, r = rem >> 3, c = rem & 7, base = 2 * pr * h;
      const double* Crow = s + (base + h + r) * PT + base;
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < h; ++k) sum += (k >= c) ? Crow[k] * s[(base + k) * PT + base + c] : 0.0;
      s[(base + c) * PT + base + h + r] = sum;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * h; idx += 256) {  // X21 = -B^-1 T
      const int pr = idx >> 6, rem = idx & 63, r = rem >> 3, c = rem & 7, base = 2 * pr * h;
      const double* Brow = s + (base + h + r) * PT + base + h;
      const double* Tcol = s + (base + c) * PT + base + h;
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < h; ++k) sum += (k <= r) ? Brow[k] * Tcol[k] : 0.0;
      s[(base + h + r) * PT + base + c] = -sum;
    }
  }
  potrf_tail(s, Linv_out, rhs, yout, k0);
  PROBE(5);
}

// Blocked form of the same kernel (COVGPU_POTRF=2): the 128 pivot steps of k_potrf_inv cost 0.47 us each — one barrier,
// one LDS round trip and a 64-FMA register update per pivot — and 40 of these kernels sit on the serial chain of every
// linear solve. Here the tile stays in LDS and is factored 16 columns at a time:
//   (a) wave 0 factors the 16x16 diagonal block in registers (row per lane, v_readlane broadcasts — no barrier inside) and
//       leaves the reciprocal pivots; the eight block inverses the recursive-doubling tail starts from are formed at
//       the end, one column per thread, all blocks at once;
//   (b) every thread solves ONE row of the 16-column panel by forward substitution against that block (broadcast LDS
//       reads; a product with the explicit inverse would be cheaper but loses the backward stability the ill-conditioned
//       reduced camera system needs — k_potrf_inv's header records that dead end);
//   (c) the trailing tiles are updated on the matrix core, C_ik -= L_ij L_kj^T, read-modify-write in LDS.
// Three barriers per 16 pivots instead of sixteen; the serial part is the diagonal blocks only.
__global__ __launch_bounds__(256) void k_potrf_inv_blk(double* __restrict__ M, size_t ld, int k0, double* __restrict__ Linv_out, int* flag,
                                                        const double* __restrict__ rhs, double* __restrict__ yout, size_t bsM, size_t bsL, size_t bsR) {
  M += (size_t)blockIdx.x * bsM; Linv_out += (size_t)blockIdx.x * bsL;
  if (rhs != nullptr) { rhs += (size_t)blockIdx.x * bsR; yout += (size_t)blockIdx.x * bsR; }
  extern __shared__ __attribute__((aligned(16))) double s[];  // [128][129] | sX [8][16][16] | sInv [128]
  constexpr int PT = kTile + 1, NB = 16, NJ = kTile / NB;
  double* sX = s + kTile * PT;
  double* sInv = sX + NJ * NB * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* Mg = M + (size_t)k0 * ld + k0;
  PROBE(0);
  {  // lower triangle in, zeros above: 16 independent loads in flight per thread (one at a time took 26 us)
    const int c = tid & 127, rh = tid >> 7;
#pragma unroll 1
    for (int r0 = 0; r0 < kTile; r0 += 32) {
      double v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int r = r0 + 2 * i + rh; v[i] = (c <= r) ? Mg[(size_t)r * ld + c] : 0.0; }
#pragma unroll
      for (int i = 0; i < 16; ++i) s[(r0 + 2 * i + rh) * PT + c] = v[i];
    }
  }
  __syncthreads();
  PROBE(1);
  for (int j = 0; j < NJ; ++j) {
    const int o = NB * j;
    const long long tq0 = PROBE_T0();
    if (wave == 0) {  // (a) diagonal block: lane r (mod 16) owns row r; lanes 16..63 compute duplicates and do not store
      // (Measured alternative, tools/potrf_probe: every lane factoring the block redundantly in its own registers as 2x2 of
      //  8x8 blocks — no cross-lane traffic at all — takes 5.9 us per block against 4.5 us here: ~1000 dependent FP64 FMAs
      //  at one wave per SIMD cost more than the 30 v_readlane per pivot they replace.)
      const int r = lane & 15;
      double x[NB], invd[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] = (c <= r) ? s[(o + r) * PT + o + c] : 0.0;
      bool bad = false;
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        double d = rdlane64c(x[c], c);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        double inv = __builtin_amdgcn_rsq(d);   // hardware estimate + two Newton steps: full double precision for the
        inv = inv * (1.5 - 0.5 * d * inv * inv);  // positive, normal-range pivots of an SPD tile
        inv = inv * (1.5 - 0.5 * d * inv * inv);
        invd[c] = inv;
        x[c] = (r == c) ? d * inv : x[c] * inv;
#pragma unroll
        for (int cc = c + 1; cc < NB; ++cc) x[cc] -= x[c] * rdlane64c(x[c], cc);  // entries above the diagonal are never read
      }
      if (bad && lane == 0) atomicOr(flag, 1);
      if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) s[(o + r) * PT + o + c] = (c <= r) ? x[c] : 0.0;
        double mine = 0.0;  // reciprocal pivot of row r
#pragma unroll
        for (int c = 0; c < NB; ++c) mine = (c == r) ? invd[c] : mine;
        sInv[o + r] = mine;
      }
    }
    __syncthreads();
    PROBE_ACC(5, tq0);
    const long long tq1 = PROBE_T0();
    {  // (b) panel rows below the block: x L_D^T = a, one row per thread
      const int row = o + NB + tid;
      if (row < kTile) {
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = s[row * PT + o + c];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          a[k] *= sInv[o + k];
#pragma unroll
          for (int c = k + 1; c < NB; ++c) a[c] -= a[k] * s[(o + c) * PT + o + k];
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) s[row * PT + o + c] = a[c];
      }
    }
    __syncthreads();
    PROBE_ACC(6, tq1);
    const long long tq2 = PROBE_T0();
    {  // (c) trailing update of tiles (i, k), j < k <= i < 8
      const int m = NJ - 1 - j, nT = m * (m + 1) / 2, fr = lane & 15, fk = lane >> 4;
      for (int t = wave; t < nT; t += 4) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tk = t - ti * (ti + 1) / 2;
        const int ri = o + NB * (1 + ti), rk = o + NB * (1 + tk);
        v4f64 acc;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg] = s[(ri + fk + 4 * rg) * PT + rk + fr];
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
          const double av = -s[(ri + fr) * PT + o + 4 * ss + fk];
          const double bv = s[(rk + fr) * PT + o + 4 * ss + fk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) s[(ri + fk + 4 * rg) * PT + rk + fr] = acc[rg];
      }
    }
    __syncthreads();
    PROBE_ACC(7, tq2);
  }
  PROBE(2);
  // L back to HBM, then the diagonal blocks are replaced by their inverses for the recursive-doubling tail
  {
    const int c = tid & 127, rh = tid >> 7;
#pragma unroll 8
    for (int r = rh; r < kTile; r += 2)
      if (c <= r) Mg[(size_t)r * ld + c] = s[r * PT + c];
  }
  __syncthreads();
  if (tid < kTile) {  // inverses of the eight 16x16 diagonal blocks, one column per thread (forward substitution on L e_c)
    const int j = tid >> 4, col = tid & 15, o = NB * j;
    double xi[NB];
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < rr; ++k) sum += s[(o + rr) * PT + o + k] * xi[k];
      xi[rr] = (rr == col) ? sInv[o + rr] : (rr > col ? -sum * sInv[o + rr] : 0.0);
    }
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) sX[(j * NB + rr) * NB + col] = xi[rr];
  }
  __syncthreads();
  for (int idx = tid; idx < NJ * NB * NB; idx += 256) {
    const int j = idx >> 8, rr = (idx >> 4) & 15, cc = idx & 15;
    s[(NB * j + rr) * PT + NB * j + cc] = sX[idx];
  }
  PROBE(3);
  potrf_tail(s, Linv_out, rhs, yout, k0);
  PROBE(5);
}

// backward substitution step for panel p:  x_p = Linv_p^T y_p ; y[cols left of the panel] -= L[panel rows, cols]^T x_p
// Block = 32 columns x 8 row groups (16 panel rows each): 8x more loads in flight than one thread per column
// (that version was latency-bound: 50 us per step), partial sums combined through LDS in a fixed order.
// Linv == nullptr: x_p is given (block-arrow solve: a border tile); ncol = number of columns to update (p*128 normally).
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ M, size_t ld, int p, const double* __restrict__ Linv,
                                                   double* __restrict__ y, double* __restrict__ x, int ncol, size_t bsM, size_t bsL, size_t bsR) {
  M += (size_t)blockIdx.y * bsM; y += (size_t)blockIdx.y * bsR; x += (size_t)blockIdx.y * bsR;  // batched form
  if (Linv != nullptr) Linv += (size_t)blockIdx.y * bsL;
  __shared__ double sx[kTile];
  __shared__ double sy[kTile];
  __shared__ double part[8][33];
  const int tid = threadIdx.x, k0 = p * kTile;
  if (Linv == nullptr) {
    if (tid < kTile) sx[tid] = x[k0 + tid];
    __syncthreads();
  } else {  // x_p = Linv^T y_p : thread (i, half) sums half of the rows j >= i
    if (tid < kTile) sy[tid] = y[k0 + tid];
    __syncthreads();
    const int i = tid & 127, half = tid >> 7;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double* Lc = Linv + i;
    const int j0 = half * 64, j1 = j0 + 64;
#pragma unroll 4
    for (int j = j0; j < j1; j += 4) {  // Linv is zero above the diagonal: no j >= i test needed
      s0 += Lc[(size_t)j * kTile] * sy[j]; s1 += Lc[(size_t)(j + 1) * kTile] * sy[j + 1];
      s2 += Lc[(size_t)(j + 2) * kTile] * sy[j + 2]; s3 += Lc[(size_t)(j + 3) * kTile] * sy[j + 3];
    }
    const double v = (s0 + s1) + (s2 + s3);
    if (half == 1) sx[i] = v;
    __syncthreads();
    if (half == 0) { sx[i] += v; }
    __syncthreads();
    if (blockIdx.x == 0 && tid < kTile) x[k0 + tid] = sx[tid];
  }
  const int cl = tid & 31, rg = tid >> 5;
  const int col = blockIdx.x * 32 + cl;
  double acc = 0.0;
  if (col < ncol) {
    const double* Lc = M + (size_t)(k0 + 16 * rg) * ld + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += Lc[(size_t)r * ld] * sx[16 * rg + r];
  }
  part[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && col < ncol) {
    double t = 0.0;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) t += part[g2][cl];
    y[col] -= t;
  }
}

// The single-workgroup potrf (133 KB LDS + 174 VGPRs x 256 threads: needs an EMPTY CU) starves for the whole duration
// of a bulk trailing update when every CU holds two bulk workgroups: 400-800 us instead of ~100 us (profiles/r01q,
// timeline in profiles/r01y_timeline_*.csv). Creating the bulk (aux) and rest-row (mid) streams with a CU mask that
// keeps N CUs per XCD out of their reach removes the starvation (potrf 104 us flat). Measured mask-bit numbering
// on this stack: bit i -> XCD i % 8, CU i / 8; bits 0..8N-1 = CUs 0..N-1 of every XCD (reserving CUs of one XCD
// only unbalances the XCD-static supertile schedule of the bulk: -10 %).
// What the timeline shows: panel P+2 cannot start before bulk(P) has finished (its tiles are written by it), so the
// factorisation runs in two regimes. While the trailing matrix is large (first ~22 of 52 big panels on the 5-agent
// map) the period is one bulk launch and the chain hides inside it — there the mask only costs (bulk 8 % slower,
// 37.5 vs 40.7 TFLOP/s); once the bulk is shorter than the ~400 us chain the period is the chain — there the mask
// helps (no starvation). The two effects cancel: factor+solve 34.3-34.4 ms with N = 1, 2, 4 against 34.1-34.8 ms
// without. The mask is therefore OPT-IN (COVGPU_CU_MASK=N).
static hipStream_t make_side_stream(int priority) {
  hipStream_t s2 = nullptr;
  const char* on = getenv("COVGPU_CU_MASK");  // N = CUs per XCD kept free for the main stream (bits 0 .. 8N-1)
  const int nres = on ? atoi(on) : 0;
  if (nres > 0 && nres <= 8) {
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = 0xFFFFFFFFu;
    for (int b = 0; b < 8 * nres; ++b) mask[b >> 5] &= ~(1u << (b & 31));
    if (hipExtStreamCreateWithCUMask(&s2, 8, mask) == hipSuccess) return s2;
  }
  (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, priority);
  return s2;
}

void CholAux::init() {
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (!head) (void)hipStreamCreateWithPriority(&head, hipStreamNonBlocking, hi);
  if (!mid) mid = make_side_stream(hi);
  if (!aux) aux = make_side_stream(lo);
  if (!ev_sb) (void)hipEventCreateWithFlags(&ev_sb, hipEventDisableTiming);
  if (!ev_cf) (void)hipEventCreateWithFlags(&ev_cf, hipEventDisableTiming);
  if (!ev_g) (void)hipEventCreateWithFlags(&ev_g, hipEventDisableTiming);
  if (!ev_z) (void)hipEventCreateWithFlags(&ev_z, hipEventDisableTiming);
  if (!ev_fill) (void)hipEventCreateWithFlags(&ev_fill, hipEventDisableTiming);
}
void CholAux::TriCache::clear() {
  for (int* p : list) if (p) (void)hipFree(p);
  list.clear(); count.clear(); key = -1;
}
void CholAux::tri_clear() {
  tri0.clear();
  for (auto& t : tri_lev) t.clear();
}
void CholAux::destroy() {
  tri_clear();
  if (mid) { (void)hipStreamDestroy(mid); mid = nullptr; }
  if (head) { (void)hipStreamDestroy(head); head = nullptr; }
  for (auto e : ev) (void)hipEventDestroy(e);
  for (auto e : prof_ev) (void)hipEventDestroy(e);
  for (auto e : panel_ev) (void)hipEventDestroy(e);
  panel_ev.clear();
  ev.clear(); prof_ev.clear();
  if (ev_sb) { (void)hipEventDestroy(ev_sb); ev_sb = nullptr; }
  if (ev_cf) { (void)hipEventDestroy(ev_cf); ev_cf = nullptr; }
  if (ev_g) { (void)hipEventDestroy(ev_g); ev_g = nullptr; }
  if (ev_z) { (void)hipEventDestroy(ev_z); ev_z = nullptr; }
  if (ev_fill) { (void)hipEventDestroy(ev_fill); ev_fill = nullptr; }
  cf_pending = false;
  if (aux) { (void)hipStreamDestroy(aux); aux = nullptr; }
}
void CholAux::mark(hipStream_t s, int tag) {
  static const bool trace_panels = getenv("COVGPU_TRACE_PANELS") != nullptr;
  if (!trace_panels) return;
  while ((int)panel_ev.size() <= panel_n) { hipEvent_t e; (void)hipEventCreate(&e); panel_ev.push_back(e); }
  if ((int)panel_tag.size() <= panel_n) panel_tag.resize(panel_n + 1);
  (void)hipEventRecord(panel_ev[panel_n], s);
  panel_tag[panel_n++] = tag;
}
// after the streams have been synchronised: accumulate the bracketed trailing-update launches
void CholAux::collect() {
  if (panel_n > 1) {  // un-profiled progression of the last solve: when did each big panel's chain start?
    fprintf(stderr, "covgpu marks [us]:");
    for (int i = 1; i < panel_n; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, panel_ev[0], panel_ev[i]) == hipSuccess) fprintf(stderr, " %d:%.0f", panel_tag[i], ms * 1e3);
    }
    fprintf(stderr, "\n");
  }
  panel_n = 0;
  if (!profile) return;
  for (size_t i = 0; i < prof_flops.size(); ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, prof_ev[2 * i], prof_ev[2 * i + 1]) != hipSuccess) continue;
    syrk_ms += ms; syrk_flops += prof_flops[i]; n_syrk++;
  }
  prof_flops.clear();
}

bool dense_panel_chain() {
  static const bool on = [] { const char* e = getenv("COVGPU_PANEL"); return e ? atoi(e) != 0 : true; }();
  return on;
}

void dense_cholesky_solve_raw(double* S, double* b, double* Linv, int* flag, int npad, hipStream_t st, CholAux& ax, int tstop, bool solve,
                              DenseBatch bt) {
  const bool panel256 = dense_panel_chain();
  const int nbt = bt.n > 0 ? bt.n : 1;
  const int T = npad / kTile;
  const size_t ld = (size_t)npad;
  const size_t lds_potrf = (size_t)kTile * (kTile + 1) * sizeof(double);
  const size_t lds_potrf_blk = lds_potrf + (size_t)(8 * 16 * 16 + kTile) * sizeof(double);
  const size_t lds_gemm = (size_t)2 * kTile * LDT * sizeof(double);
  static std::once_flag attr_once;  // (the pose-graph solve calls this from several host threads at once)
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_potrf);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv_blk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_potrf_blk);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_TRSM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK_TRI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK_RECT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
  });
  ax.init();
  const int NP = (T + 1) / 2;  // big panels of two tile columns
  // events per big panel: H rows-h done | B bulk done | C rows-r done | 1 potrf(t0) | 2 X(t0+1,t0) | 3 potrf(t0+1) | Rc next diagonal updated
  while ((int)ax.ev.size() < 8 * (NP + 1)) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); ax.ev.push_back(e); }
  // (a solve may factorise several matrices — arrow blocks, then the border system: launches accumulate until collect())
  if (ax.profile) while (ax.prof_ev.size() < 2 * (ax.prof_flops.size() + (size_t)NP)) { hipEvent_t e; (void)hipEventCreate(&e); ax.prof_ev.push_back(e); }
  hipEvent_t* eH = ax.ev.data();
  hipEvent_t* eB = eH + (NP + 1);
  hipEvent_t* eC = eB + (NP + 1);
  hipEvent_t* e1 = eC + (NP + 1);
  hipEvent_t* e2 = e1 + (NP + 1);
  hipEvent_t* e3 = e2 + (NP + 1);
  hipEvent_t* eRc = e3 + (NP + 1);
  hipEvent_t* eHp = eRc + (NP + 1);  // rows h updated with column t0 (their last TRSM then runs on the chain's own stream)

  static const int potrf_kind = [] { const char* e = getenv("COVGPU_POTRF"); return e ? atoi(e) : 2; }();  // 1: per-pivot kernel, 2: blocked (default)
  auto potrf = [&](int t) {
    if (potrf_kind == 1)
      hipLaunchKernelGGL(k_potrf_inv, dim3(nbt), dim3(256), lds_potrf, st, S, ld, t * kTile, Linv + (size_t)t * kTile * kTile, flag, b, b + npad, bt.sM, bt.sL, bt.sR);
    else
      hipLaunchKernelGGL(k_potrf_inv_blk, dim3(nbt), dim3(256), lds_potrf_blk, st, S, ld, t * kTile, Linv + (size_t)t * kTile * kTile, flag, b, b + npad, bt.sM, bt.sL, bt.sR);
  };
  // rows [r0, r1) of tile column t:  A <- A Linv_t^T
  // quad: four workgroups per tile (head launches on the serial chain)
  auto trsm = [&](int t, int r0, int r1, hipStream_t s2, bool quad) {
    if (r1 <= r0) return;
    GemmArgs g{S, ld, t * kTile, kTile, r0 * kTile, 0, t * kTile, r1 - r0, Linv + (size_t)t * kTile * kTile, b, b + npad, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab};
    if (quad) hipLaunchKernelGGL((k_gemm_abt_q<MODE_TRSM, 32, kTile>), dim3(r1 - r0, 1, 4 * nbt), dim3(256), (size_t)(32 + kTile) * (KCQ + 1) * sizeof(double), s2, g);
    else hipLaunchKernelGGL(k_gemm_abt<MODE_TRSM>, dim3(r1 - r0, 1, nbt), dim3(256), lds_gemm, s2, g);
  };
  // C tiles (rows [r0, r1), tile columns [tc0, tc0+ntc)) -= A[rows, K] A[tc.., K]^T, K = tiles kt0.. (KD columns); lower part only
  auto rect = [&](int r0, int r1, int tc0, int ntc, int kt0, int KD, hipStream_t s2, bool quad) {
    if (r1 <= r0 || ntc <= 0 || KD <= 0) return;
    GemmArgs g{S, ld, kt0 * kTile, KD, r0 * kTile, tc0 * kTile, tc0 * kTile, r1 - r0, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab};
    if (quad) hipLaunchKernelGGL((k_gemm_abt_q<MODE_SYRK_RECT, 64, 64>), dim3(ntc, r1 - r0, 4 * nbt), dim3(256), (size_t)(64 + 64) * (KCQ + 1) * sizeof(double), s2, g);
    else hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_RECT>, dim3(ntc, r1 - r0, nbt), dim3(256), lds_gemm, s2, g);
  };

  // Four streams (timeline analysis in profiles/r01y_timeline_*.csv: once the trailing matrix is small the period of
  // the factorisation is the serial chain potrf -> trsm -> rect -> potrf ..., so nothing else may sit on it):
  //   M  (= st) the critical chain only: the panel's 2x2 diagonal tiles, and the look-ahead update of the NEXT
  //             panel's 2x2 diagonal tiles;
  //   H  rows h = the next panel's two tile rows (t0+2, t0+3): their TRSMs / updates trail the chain by one kernel;
  //   R  rows r = everything below (t0+4 ..): full-tile kernels, needed one panel later;
  //   B  the bulk rank-256 trailing update (triangle from tile t0+4).
  hipStream_t M = st, H = ax.head, R = ax.mid, B = ax.aux;
  auto wait = [](hipStream_t s2, hipEvent_t e) { (void)hipStreamWaitEvent(s2, e, 0); };
  // Partial factorisation (tstop >= 0, even): eliminate tile columns [0, tstop) only; the trailing block then holds
  // its Schur complement (and, with the forward substitution riding along, b's trailing part the reduced right-hand
  // side). Used by the block-arrow pose-graph solve (k_pgo.hip).
  const int Pstop = (tstop >= 0 && tstop < T) ? tstop / 2 : NP;
  // K range of big panel Pp's rank update: its columns beyond the batch's largest real interior order are identity padding
  // with zeros below (DenseBatch::own_max) — multiples of 32 (chunk of the quarter-tile kernels); 0: nothing to apply
  auto kd = [&](int Pp) {
    const int full = std::min(2, T - 2 * Pp) * kTile;
    if (bt.own_max <= 0) return full;
    return std::max(0, std::min(full, ((bt.own_max - 2 * Pp * kTile + 31) / 32) * 32));
  };
  if (bt.tri_slot >= (int)ax.tri_lev.size()) ax.tri_lev.resize(bt.tri_slot + 1);
  CholAux::TriCache& tc = bt.tri_slot >= 0 ? ax.tri_lev[bt.tri_slot] : ax.tri0;
  if (bt.live_h != nullptr && tc.key != T * 4096 + nbt) {  // live-tile lists of every panel's bulk update (static per problem)
    tc.clear();
    tc.key = T * 4096 + nbt;
    tc.list.assign(NP, nullptr); tc.count.assign(NP, 0);
    for (int P = 0; P < NP && P < Pstop; ++P) {
      // (the LAST panel of a partial factorisation applies its whole trailing update in one launch: triangle from t0 + 2)
      const int t0 = 2 * P, tb = (panel256 && Pstop < NP && P == Pstop - 1) ? t0 + 2 : t0 + 4;
      if (tb >= T) break;
      std::vector<int> q[8];  // per-XCD queues; whole 8x8 supertiles of one batch go to the currently shortest queue
      for (int a = 0; a < nbt; ++a) {
        const int nI = bt.live_h[2 * a], nO = bt.live_h[2 * a + 1];
        if (t0 >= nI) continue;
        auto live = [&](int t) { return t < nI || (t >= bt.tI && t - bt.tI < nO); };
        const int nt = T - tb, Ts = (nt + 7) / 8;
        for (int si = 0; si < Ts; ++si)
          for (int sj = 0; sj <= si; ++sj) {
            std::vector<int> grp;
            for (int i = 8 * si; i < std::min(nt, 8 * si + 8); ++i)
              for (int j = 8 * sj; j < std::min(i + 1, 8 * sj + 8); ++j)
                if (live(tb + i) && live(tb + j)) grp.push_back((a << 20) | (i << 10) | j);
            if (grp.empty()) continue;
            int best = 0;
            for (int x = 1; x < 8; ++x) if (q[x].size() < q[best].size()) best = x;
            q[best].insert(q[best].end(), grp.begin(), grp.end());
          }
      }
      size_t longest = 0;
      for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
      if (longest == 0) continue;
      std::vector<int> lst(8 * longest, -1);
      for (int x = 0; x < 8; ++x) for (size_t k = 0; k < q[x].size(); ++k) lst[8 * k + x] = q[x][k];
      int* d = nullptr;
      if (hipMalloc((void**)&d, lst.size() * sizeof(int)) != hipSuccess) { tc.clear(); break; }
      (void)hipMemcpy(d, lst.data(), lst.size() * sizeof(int), hipMemcpyHostToDevice);
      tc.list[P] = d; tc.count[P] = (int)lst.size();
    }
  }
  int Plast = NP - 1;
  for (int P = 0; P < NP; ++P) {
    const int t0 = 2 * P, w = (T - t0 >= 2) ? 2 : 1;
    const int h0 = (t0 + 2 < T) ? t0 + 2 : T, h1 = (t0 + 4 < T) ? t0 + 4 : T;  // rows h = [h0, h1), rows r = [h1, T)
    // ---- look-ahead part of SYRK(P-1) (K = the 256 columns of panel P-1) on this panel's two tile columns; the
    //      2x2 diagonal part was enqueued on M at the end of the previous iteration
    if (P > 0) {
      if (h1 > h0) {
        wait(H, eC[P - 1]);                    // L rows h0.. were rest rows of panel P-1
        wait(H, eH[P - 1]);                    // B operand: L rows t0, t0+1 — their last TRSM ran on the chain's stream
        if (P >= 2) wait(H, eB[P - 2]);        // bulk(P-2) was the previous writer of these tiles
        rect(h0, h1, t0, w, t0 - 2, kd(P - 1), H, true);
        if (panel256) (void)hipEventRecord(eHp[P], H);
      }
      if (T > h1) {
        wait(R, eH[P - 1]);                    // B operand: L rows t0, t0+1 (rows h of panel P-1)
        if (P >= 2) wait(R, eB[P - 2]);
        rect(h1, T, t0, w, t0 - 2, kd(P - 1), R, false);
        if (panel256) (void)hipEventRecord(e2[P], R);  // (e2 is free in the 256-column chain: rows r carry panel P-1's update)
      }
    }
    if (P == Pstop) {  // only the look-ahead updates of the last eliminated panel; nothing of this panel is factored
      (void)hipEventRecord(eH[P], H);
      (void)hipEventRecord(eC[P], R);
      (void)hipEventRecord(eB[P], B);
      Plast = P;
      break;
    }
    // ---- M: critical chain
    ax.mark(M, P);
    if (panel256 && Pstop < NP && P == Pstop - 1) {
      // LAST panel of a partial factorisation (every multifrontal front, every arrow block): nothing is factored after it, so
      // the look-ahead split of its trailing update (next panel's rows / diagonal / rest rows / bulk: four launches on three
      // streams, ~100 us of event hops per front level) buys nothing — factor, solve ALL rows below, update the WHOLE trailing
      // triangle, three dependent launches on the chain's own stream.
      const int nbp = bt.own_max > 0 ? std::max(0, std::min(8 * w, (bt.own_max - t0 * kTile + 15) / 16)) : -1;
      launch_potrf_panel(S, ld, t0, w, Linv, flag, b, npad, nbt, bt.sM, bt.sL, bt.sR, M, bt.tab, nbp);
      (void)hipEventRecord(e1[P], M);
      if (T > h0) {
        if (P > 0) { if (h1 > h0) wait(M, eHp[P]); if (T > h1) wait(M, e2[P]); }
        launch_trsm_sub(S, ld, t0, w, h0, T, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, M, true, bt.tab, nbp);
        if (P >= 1) wait(M, eB[P - 1]);  // bulk(P-1) was the previous writer of the trailing tiles
        const int tb = h0, nt = T - tb;
        if (kd(P) > 0) {
          const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
          GemmArgs g{S, ld, t0 * kTile, kd(P), tb * kTile, tb * kTile, tb * kTile, nt, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab};
          const bool listed = bt.live_h != nullptr && P < (int)tc.list.size() && tc.list[P] != nullptr;
          if (listed) g.tri = tc.list[P];
          if (!listed || tc.count[P] > 0) {
            double pairs = 0.0;
            for (int a = 0; a < nbt; ++a) {
              if (bt.live_h == nullptr) { pairs += (double)nt * (nt + 1) / 2; continue; }
              const int nI = bt.live_h[2 * a], nO = bt.live_h[2 * a + 1];
              if (t0 >= nI) continue;
              int nl = 0;
              for (int t = tb; t < T; ++t) nl += (t < nI || (t >= bt.tI && t - bt.tI < nO)) ? 1 : 0;
              pairs += (double)nl * (nl + 1) / 2;
            }
            if (ax.profile) (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size()], M);
            if (listed) hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(tc.count[P], 1), dim3(256), lds_gemm, M, g);
            else hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(nblk, nbt), dim3(256), lds_gemm, M, g);
            if (ax.profile) {
              (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size() + 1], M);
              ax.prof_flops.push_back(pairs * 2.0 * kTile * kTile * kd(P));
            }
          }
        }
      }
      (void)hipEventRecord(eH[P], M); (void)hipEventRecord(eC[P], M); (void)hipEventRecord(eB[P], M);
      Plast = P;
      break;
    }
    if (panel256) {
      // 256-column chain (k_panel.hip): one workgroup factors the whole diagonal block, rows h follow on the same stream by
      // block substitution, rows r on theirs — three dependent launches per panel instead of six
      const int nbp = bt.own_max > 0 ? std::max(0, std::min(8 * w, (bt.own_max - t0 * kTile + 15) / 16)) : -1;
      launch_potrf_panel(S, ld, t0, w, Linv, flag, b, npad, nbt, bt.sM, bt.sL, bt.sR, M, bt.tab, nbp);
      (void)hipEventRecord(e1[P], M);
      if (h1 > h0) {
        if (P > 0) wait(M, eHp[P]);            // rows h carry the look-ahead update of panel P-1 (stream H, above)
        launch_trsm_sub(S, ld, t0, w, h0, h1, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, M, true, bt.tab, nbp);
      }
      (void)hipEventRecord(eH[P], M);
      if (T > h1) {
        wait(R, e1[P]);
        launch_trsm_sub(S, ld, t0, w, h1, T, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, R, false, bt.tab, nbp);
      }
      (void)hipEventRecord(eC[P], R);
    } else {
      potrf(t0);
      (void)hipEventRecord(e1[P], M);
      if (w == 2) {
        trsm(t0, t0 + 1, t0 + 2, M, true);
        (void)hipEventRecord(e2[P], M);
        rect(t0 + 1, t0 + 2, t0 + 1, 1, t0, kTile, M, true);   // rank-128 update of the second diagonal tile
        potrf(t0 + 1);
        (void)hipEventRecord(e3[P], M);
      }
      // ---- H: rows h
      if (h1 > h0) {
        wait(H, e1[P]);
        trsm(t0, h0, h1, H, true);
        if (w == 2) {
          wait(H, e2[P]);                        // X(t0+1, t0)
          rect(h0, h1, t0 + 1, 1, t0, kTile, H, true);
          // the LAST kernel of rows h — their TRSM against potrf(t0+1) — gates the next diagonal update: it runs on the chain's
          // stream right behind that potrf instead of paying a cross-stream event hop (~10 us per big panel un-profiled)
          (void)hipEventRecord(eHp[P], H);
          wait(M, eHp[P]);
          trsm(t0 + 1, h0, h1, M, true);
        }
      }
      (void)hipEventRecord(eH[P], (h1 > h0 && w == 2) ? M : H);
      // ---- R: rows r
      if (T > h1) {
        wait(R, e1[P]);
        trsm(t0, h1, T, R, false);
        if (w == 2) {
          wait(R, e2[P]);
          rect(h1, T, t0 + 1, 1, t0, kTile, R, false);
          wait(R, e3[P]);
          trsm(t0 + 1, h1, T, R, false);
        }
      }
      (void)hipEventRecord(eC[P], R);
    }
    // ---- M: look-ahead part of SYRK(P) on the next panel's 2x2 diagonal tiles
    if (P + 1 < NP) {
      const int u0 = t0 + 2, uw = (T - u0 >= 2) ? 2 : 1;
      wait(M, eH[P]);                          // L rows u0, u0+1
      if (P >= 1) wait(M, eB[P - 1]);          // bulk(P-1) was the previous writer of these tiles
      rect(u0, u0 + uw, u0, uw, t0, kd(P), M, true);
      (void)hipEventRecord(eRc[P + 1], M);
    }
    // ---- B: bulk of SYRK(P), triangle starting two tile columns further (those belong to the look-ahead)
    const int tb = t0 + 4, nt = T - tb;
    wait(B, eC[P]);
    // A bulk launched at the same instant as the next diagonal update takes every workgroup slot first and the
    // chain waits ~75 us for the first round of tiles to retire: the bulk starts after that small kernel.
    if (P + 1 < NP) wait(B, eRc[P + 1]);
    if (nt > 0 && kd(P) > 0) {
      const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
      GemmArgs g{S, ld, t0 * kTile, kd(P), tb * kTile, tb * kTile, tb * kTile, nt, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab};
      // arrow buffers: the live tiles of this panel's update as an explicit, XCD-balanced list (built once per problem).
      // The implicit triangle grid x batch launched ~2.6k workgroups of which ~400 did work, with every batch's first
      // supertile on XCD 0: 17 TFLOP/s.
      const bool listed = bt.live_h != nullptr && P < (int)tc.list.size() && tc.list[P] != nullptr;
      if (listed) g.tri = tc.list[P];
      double pairs = 0.0;  // tile pairs that do work
      for (int a = 0; a < nbt; ++a) {
        if (bt.live_h == nullptr) { pairs += (double)nt * (nt + 1) / 2; continue; }
        const int nI = bt.live_h[2 * a], nO = bt.live_h[2 * a + 1];
        if (t0 >= nI) continue;
        int nl = 0;
        for (int t = tb; t < T; ++t) nl += (t < nI || (t >= bt.tI && t - bt.tI < nO)) ? 1 : 0;
        pairs += (double)nl * (nl + 1) / 2;
      }
      if (!listed || tc.count[P] > 0) {
        if (ax.profile) (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size()], B);
        if (listed) hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(tc.count[P], 1), dim3(256), lds_gemm, B, g);
        else hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(nblk, nbt), dim3(256), lds_gemm, B, g);
        if (ax.profile) {
          (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size() + 1], B);
          ax.prof_flops.push_back(pairs * 2.0 * kTile * kTile * kd(P));
        }
      }
    }
    (void)hipEventRecord(eB[P], B);
  }
  wait(M, eB[Plast]);
  if (Plast >= 1) wait(M, eB[Plast - 1]);
  wait(M, eC[Plast]);
  wait(M, eH[Plast]);
  if (!solve) return;
  // y = L^-1 b was formed along the way (potrf: y_p = L_pp^-1 b_p; every TRSM: b[rows] -= L[rows, p] y_p) and lives
  // in b[npad .. 2 npad). Remaining: L^T x = y.
  dense_backward_solve(S, b, Linv, npad, st, T, T, bt);
}

// L^T x = y for the factored tile columns [0, tfact) of an npad-order matrix; for tile rows p in [tfact, tend) x_p is
// GIVEN (already in b[p*128 ..]) and only its contribution L[rows p, cols < tfact*128]^T x_p is taken out of y.
void dense_backward_solve(double* S, double* b, double* Linv, int npad, hipStream_t st, int tfact, int tend, DenseBatch bt) {
  const int nbt = bt.n > 0 ? bt.n : 1;
  const size_t ld = (size_t)npad;
  if (dense_panel_chain() && tend > tfact) {  // all given rows in one launch (k_panel.hip)
    launch_bwd_given(S, ld, tfact * kTile, tend * kTile, b + npad, b, tfact * kTile, nbt, bt.sM, bt.sR, st, bt.tab, bt.live, bt.tI, bt.xfer);
    tend = tfact;
  }
  if (bt.own_max > 0 && tend <= tfact) tend = std::min(tend, (bt.own_max + kTile - 1) / kTile);  // all-padding interior tiles: x = 0
  for (int p = tend - 1; p >= 0; --p) {
    const bool given = p >= tfact;
    const int ncol = given ? tfact * kTile : p * kTile;
    const int nb = (ncol + 31) / 32;
    if (given && nb == 0) continue;
    if (dense_panel_chain())
      launch_bwd_step_sub(S, ld, p, given ? nullptr : Linv + (size_t)p * kTile * kTile, b + npad, b, ncol, nb > 0 ? nb : 1, nbt, bt.sM, bt.sL, bt.sR, st, bt.tab,
                          bt.live, bt.tI, p == 0 ? bt.xfer : BwdXfer());
    else
      hipLaunchKernelGGL(k_bwd_step, dim3(nb > 0 ? nb : 1, nbt), dim3(256), 0, st, S, ld, p, given ? nullptr : Linv + (size_t)p * kTile * kTile,
                         b + npad, b, ncol, bt.sM, bt.sL, bt.sR);
  }
}

}  // namespace covgpu
