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
constexpr int kLdsPad = 2;   // LDS pitch = K chunk + 2 doubles: the fragment reads (lane -> row fr, column kk + fk) of a half-wave then hit 32 distinct bank pairs.
                            // Rounds 1-4 used + 1 ("odd pitch" — odd in DOUBLES, i.e. 34 / 66 dwords): two-way conflicts, SQ_LDS_BANK_CONFLICT = 25 % of
                            // the LDS cycles of the quarter-tile kernel (profiles/r05*_pmc_gemm_*.csv)
constexpr int LDT = KC + kLdsPad;   // LDS pitch (doubles) of the full-tile kernels
#ifndef COVGPU_PFF
#define COVGPU_PFF 1
#endif
#ifndef COVGPU_PFQ
#define COVGPU_PFQ 2   // round 4: 4 (+2 % on a map whose trailing updates were ~300 workgroups: one per CU). Round 5, corrected map (~1 000 workgroups per
                       // launch): two chunks in flight are 138 VGPRs instead of 210 — three workgroups per CU instead of two: 233.9 / 233.5 against 231.8 / 231.2 it/s
#endif
#ifndef COVGPU_PFR
#define COVGPU_PFR 2   // the 64x64 RECT form (a dozen workgroups on the serial chain: the next panel's diagonal block): chunks in flight
#endif
constexpr int PFF = COVGPU_PFF, PFQ = COVGPU_PFQ, PFR = COVGPU_PFR;   // chunks in flight in registers: full tiles, quarter forms

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
  // [batch] real interior order of every matrix (nullptr: KD applies to all): the K range stops at a front's own last real column
  const int* own;
  // SYRK_TRI, first panel (kcol0 == 0) of multifrontal fronts: a border x border tile whose map entry is 0 was NOT cleared (no child
  // adds into it, DenseBatch::beta0): the accumulator starts from zero instead of from memory
  const int* beta0_off; const int* beta0;
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
// workgroup barrier that orders LDS traffic only (no s_waitcnt vmcnt(0): global loads stay in flight across it)
COV_DEV void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE, int TSA, int TSB, int KC, int PF>
COV_DEV void gemm_abt_body(const GemmArgs& g) {
  constexpr int LDT = KC + kLdsPad;
  static_assert((TSA == kTile && TSB == kTile) || (MODE != MODE_TRSM && TSA == 64 && TSB == 64) ||
                (MODE == MODE_TRSM && TSA == 32 && TSB == kTile), "quarter forms: RECT / TRI 64x64, TRSM 32x128");
  constexpr int WGR = (TSA == 32) ? 1 : 2, WGC = 4 / WGR;  // wave grid
  constexpr int WTR = TSA / WGR, WTC = TSB / WGC;           // wave tile
  constexpr int NMR = WTR / 16, NMC = WTC / 16;             // MFMA tiles per wave
  constexpr int ZQ = (TSA == kTile && TSB == kTile) ? 1 : 4;  // quarter index and batch index share blockIdx.z
  if (ZQ == 4 && MODE != MODE_SYRK_TRI) __builtin_amdgcn_s_setprio(3);  // RECT / TRSM quarter forms run on the serial chain only, beside bulk waves: issue priority over them
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
    if (ZQ == 4 && ti == tj && qc > qr) return;   // quadrant above the diagonal
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
  const int kbeg = 0;
  int kend = g.KD;
  if (MODE != MODE_TRSM && g.own != nullptr) {
    const int real = g.own[batch] - g.kcol0;
    if (real <= 0) return;   // (wave-uniform) this front has no real column in the panel: A = B = 0
    kend = min(kend, ((real + KC - 1) / KC) * KC);
  }
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
  // PF chunks are in flight in registers (stage = chunk index mod PF: the K loop is unrolled by PF). With one chunk ahead the quarter
  // forms ran at memory latency — a chunk is 0.85 us of matrix work, a load under load 2-3 us, and a launch of ~300 workgroups leaves one
  // or two per CU: nothing else to run meanwhile.
  double2 pa[PF][NPA], pb[PF][NPB];
  auto gload = [&](int kc, int st) {
    const char* Ak = Ag + (size_t)kc * sizeof(double);  // uniform
    const char* Bk = Bg + (size_t)kc * sizeof(double);
#pragma unroll
    for (int it = 0; it < NPA; ++it) pa[st][it] = *reinterpret_cast<const double2*>(Ak + (size_t)(RPP * it) * ldab + offa);
#pragma unroll
    for (int it = 0; it < NPB; ++it) pb[st][it] = *reinterpret_cast<const double2*>(Bk + (size_t)(RPP * it) * ldbb + offb);
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) if (kbeg + u * KC < kend) gload(kbeg + u * KC, u);
  const int fr = lane & 15, fk = lane >> 4;
  // f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  const unsigned offc = (unsigned)(wr * WTR + fk) * ldab + (unsigned)(wc * WTC + fr) * 8u;  // per lane; the rest is uniform
  bool fresh = false;   // (wave-uniform) the tile was not cleared: start from zero
  if (MODE == MODE_SYRK_TRI && g.beta0 != nullptr && g.kcol0 == 0) {
    const int a = g.ra0 / kTile + ti - g.tI, b = g.cc0 / kTile + tj - g.tI;
    fresh = b >= 0 && a >= b && g.beta0[g.beta0_off[batch] + a * (a + 1) / 2 + b] == 0;
  }
  v4f64 acc[NMR][NMC];
#pragma unroll
  for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const unsigned rowoff = offc + (unsigned)(tm * 16 + 4 * rg) * ldab;  // one VGPR per row; tn goes into the immediate
#pragma unroll
      for (int tn = 0; tn < NMC; ++tn) {
        if (MODE == MODE_TRSM || fresh) acc[tm][tn][rg] = 0.0;
        else acc[tm][tn][rg] = *reinterpret_cast<const double*>(Cg + rowoff + tn * 128);
      }
    }
  const double sgn = (MODE == MODE_TRSM) ? 1.0 : -1.0;  // SYRK: acc = C - A B^T through a negated A fragment
  for (int kc0 = kbeg; kc0 < kend; kc0 += PF * KC) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kc = kc0 + u * KC;
      if (kc >= kend) break;
      lds_only_barrier();  // previous chunk fully consumed (LDS traffic only: __syncthreads() would also wait for the chunks in flight)
#pragma unroll
      for (int it = 0; it < NPA; ++it) { sA[rbase + RPP * it][c2] = sgn * pa[u][it].x; sA[rbase + RPP * it][c2 + 1] = sgn * pa[u][it].y; }
#pragma unroll
      for (int it = 0; it < NPB; ++it) { sB[rbase + RPP * it][c2] = pb[u][it].x; sB[rbase + RPP * it][c2 + 1] = pb[u][it].y; }
      lds_only_barrier();
      if (kc + PF * KC < kend) gload(kc + PF * KC, u);  // refill this stage while the matrix cores work
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
__global__ __launch_bounds__(256, 2) void k_gemm_abt(GemmArgs g) { gemm_abt_body<MODE, kTile, kTile, KC, PFF>(g); }
template <int MODE, int TSA, int TSB>
__global__ __launch_bounds__(256, 2) void k_gemm_abt_q(GemmArgs g) { gemm_abt_body<MODE, TSA, TSB, KCQ, (MODE == MODE_SYRK_RECT ? PFR : PFQ)>(g); }

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
  if (!ev_zero) (void)hipEventCreateWithFlags(&ev_zero, hipEventDisableTiming);
  if (!ev_lin) (void)hipEventCreateWithFlags(&ev_lin, hipEventDisableTiming);
  if (!ev_kf) (void)hipEventCreateWithFlags(&ev_kf, hipEventDisableTiming);
  if (!ev_fill) (void)hipEventCreateWithFlags(&ev_fill, hipEventDisableTiming);
  if (!ev_xb) (void)hipEventCreateWithFlags(&ev_xb, hipEventDisableTiming);
  if (!ev_xa) (void)hipEventCreateWithFlags(&ev_xa, hipEventDisableTiming);
  if (!bwd_cnt && hipMalloc((void**)&bwd_cnt, 65536 * sizeof(int)) == hipSuccess) (void)hipMemset(bwd_cnt, 0, 65536 * sizeof(int));
  // (the "gave up" word and its host mirror serve the gates and the pipelined backward substitution alike)
  if (!gate_dead && hipMalloc((void**)&gate_dead, sizeof(int)) == hipSuccess) (void)hipMemset(gate_dead, 0, sizeof(int));
  if (!gate_dead_h && hipHostMalloc((void**)&gate_dead_h, 4 * sizeof(int), hipHostMallocDefault) == hipSuccess) gate_dead_h[0] = gate_dead_h[1] = gate_dead_h[2] = gate_dead_h[3] = 0;
  if (const char* e = getenv("COVGPU_GATE_TIMEOUT_S")) gate_timeout_s = std::max(getenv("COVGPU_GATE_TIMEOUT_MIN") ? atof(getenv("COVGPU_GATE_TIMEOUT_MIN")) : 0.01, atof(e));   // (the tests of the fallback set it below a kernel's duration)
  if (!gate_flags && !gates_broken) {
    static const bool want = getenv("COVGPU_GATES") == nullptr || atoi(getenv("COVGPU_GATES")) != 0;
    if (want && gate_dead != nullptr && gate_dead_h != nullptr && hipMalloc((void**)&gate_flags, kGateSlots * sizeof(long long)) == hipSuccess) {
      (void)hipMemset(gate_flags, 0, kGateSlots * sizeof(long long));
      (void)hipDeviceSynchronize();   // (once per context: the fills are complete before the first gate polls)
      gates_on = true;
      if (getenv("COVGPU_GATE_LOG") && hipMalloc((void**)&gate_log, 2 * (size_t)kGateLogMax * sizeof(long long)) != hipSuccess) gate_log = nullptr;
    } else gates_broken = true;   // (events)
  }
}
// ---- device-flag ordering between the streams of a context (common.hpp: CholAux::record / wait)
#ifndef COVGPU_SIGNAL_RMW
#define COVGPU_SIGNAL_RMW 0
#endif
__global__ void k_signal(long long* flag, long long seq, long long* log) {
  // (a plain agent-scope store: an event is recorded on ONE stream, so the numbers of a slot arrive in order; a returning read-modify-write here
  //  made the boundary behind the signal 4-5 us instead of ~1: covgpu gate log, round 6)
  if (threadIdx.x == 0) {
    if (COVGPU_SIGNAL_RMW) __hip_atomic_fetch_max(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (log) log[0] = wall_clock64();
  }
}
__global__ void k_signal2(DevSignal a, DevSignal b) {
  if (threadIdx.x != 0) return;
  if (a.flag != nullptr) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b.flag != nullptr) __hip_atomic_store(b.flag, b.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct GateArgs { const long long* f[4]; long long s[4]; int n; int* dead; int* dead_h; long long limit; long long* log; long long* pf[2]; long long ps[2]; int np; const long long* base; };
__global__ void k_gate(GateArgs g) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < g.np; ++i) __hip_atomic_store(g.pf[i], g.ps[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // CholAux::sync: records first
  const long long t0 = wall_clock64();   // 100 MHz
  struct Stamp { long long* log; long long t0; __device__ ~Stamp() { if (log) { log[0] = t0; log[1] = wall_clock64(); } } } stamp{g.log, t0};
  for (int i = 0; i < g.n; ++i) {
    unsigned spins = 0;
    while (__hip_atomic_load(g.f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.s[i]) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 127u) == 0) {
        if (__hip_atomic_load(g.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        if (wall_clock64() - t0 > g.limit) {
          __hip_atomic_store(g.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // which wait gave up (printed with the host's warning): slot, the number awaited, the number found
          __hip_atomic_store(g.dead_h + 1, (int)(g.f[i] - g.base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(g.dead_h + 2, (int)g.s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(g.dead_h + 3, (int)__hip_atomic_load(g.f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(g.dead_h, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          return;
        }
      }
    }
  }
}
int CholAux::gate_slot_of(hipEvent_t e, hipStream_t s) {
  const std::pair<hipEvent_t, hipStream_t> key(e, s);
  auto it = gate_slot_es.find(key);
  int slot;
  if (it != gate_slot_es.end()) slot = it->second;
  else {
    slot = (int)gate_slot_es.size();
    if (slot >= kGateSlots) return -1;   // (never on the shipped maps: 9 events per panel of the widest level, at most two streams each)
    gate_slot_es.emplace(key, slot);
    if ((int)gate_seq.size() <= slot) gate_seq.resize(slot + 1, 0);
  }
  gate_slot[e] = slot;
  return slot;
}
void CholAux::record(hipEvent_t e, hipStream_t s, int tag) {
  if (!gates_on) { (void)hipEventRecord(e, s); return; }
  const int slot = gate_slot_of(e, s);
  if (slot < 0) { gate_slot.erase(e); (void)hipEventRecord(e, s); return; }
  gate_seq[slot] = ++gate_counter;
  long long* lg = nullptr;
  if ((int)gate_tag_of_slot.size() <= slot) gate_tag_of_slot.resize(slot + 1, 0);
  gate_tag_of_slot[slot] = tag;
  if (gate_log != nullptr && gate_log_n < kGateLogMax) { lg = gate_log + 2 * (size_t)gate_log_n++; gate_log_tag.push_back(tag); gate_log_kind.push_back('S'); }
  hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, s, gate_flags + slot, gate_seq[slot], lg);
  ++gate_signals;
}
DevSignal CholAux::publish_handle(hipEvent_t e, hipStream_t s, int tag) {
  DevSignal d;
  if (!gates_on) return d;
  const int slot = gate_slot_of(e, s);
  if (slot < 0) { gate_slot.erase(e); return d; }
  gate_seq[slot] = ++gate_counter;
  if ((int)gate_tag_of_slot.size() <= slot) gate_tag_of_slot.resize(slot + 1, 0);
  gate_tag_of_slot[slot] = tag;
  d.flag = gate_flags + slot; d.seq = gate_seq[slot];
  ++gate_signals;
  return d;
}
void CholAux::record_handle(DevSignal d, hipStream_t s) { if (d.flag != nullptr) hipLaunchKernelGGL(k_signal2, dim3(1), dim3(64), 0, s, d, DevSignal()); }
void CholAux::wait(hipStream_t s, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3) { sync(s, nullptr, 0, nullptr, 0, e0, e1, e2, e3); }
void CholAux::sync(hipStream_t s, hipEvent_t r0, int tag0, hipEvent_t r1, int tag1, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3) {
  hipEvent_t es[4] = {e0, e1, e2, e3};
  static const bool merged = getenv("COVGPU_GATE_MERGE") == nullptr || atoi(getenv("COVGPU_GATE_MERGE")) != 0;
  if (!gates_on || !merged) {
    if (r0 != nullptr) record(r0, s, tag0);
    if (r1 != nullptr) record(r1, s, tag1);
    if (!gates_on) { for (hipEvent_t e : es) if (e != nullptr) (void)hipStreamWaitEvent(s, e, 0); return; }
    r0 = r1 = nullptr;
  }
  GateArgs g; g.n = 0; g.dead = gate_dead; g.dead_h = gate_dead_h; g.limit = (long long)(gate_timeout_s * 1e8); g.log = nullptr; g.np = 0; g.base = gate_flags;
  g.pf[0] = g.pf[1] = nullptr; g.ps[0] = g.ps[1] = 0;
  {
    hipEvent_t rs[2] = {r0, r1}; const int tags[2] = {tag0, tag1};
    for (int i = 0; i < 2; ++i) {
      if (rs[i] == nullptr) continue;
      const int slot = gate_slot_of(rs[i], s);
      if (slot < 0) { gate_slot.erase(rs[i]); (void)hipEventRecord(rs[i], s); continue; }
      gate_seq[slot] = ++gate_counter;
      if ((int)gate_tag_of_slot.size() <= slot) gate_tag_of_slot.resize(slot + 1, 0);
      gate_tag_of_slot[slot] = tags[i];
      g.pf[g.np] = gate_flags + slot; g.ps[g.np] = gate_seq[slot]; ++g.np; ++gate_signals;
    }
  }
  int first_tag = g.np > 0 ? -(r0 != nullptr ? tag0 : tag1) : 0;   // (gate log: a negative tag marks a launch that also publishes)
  for (hipEvent_t e : es) {
    if (e == nullptr) continue;
    auto it = gate_slot.find(e);
    if (it == gate_slot.end()) { (void)hipStreamWaitEvent(s, e, 0); continue; }   // never recorded through a flag: whatever HIP knows of it
    if (gate_seq[it->second] == 0) continue;
    if (g.n == 0 && g.np == 0 && it->second < (int)gate_tag_of_slot.size()) first_tag = gate_tag_of_slot[it->second];
    g.f[g.n] = gate_flags + it->second; g.s[g.n] = gate_seq[it->second]; ++g.n;
  }
  if (g.n == 0 && g.np == 0) return;
  if (gate_log != nullptr && gate_log_n < kGateLogMax) { g.log = gate_log + 2 * (size_t)gate_log_n++; gate_log_tag.push_back(first_tag); gate_log_kind.push_back('G'); }
  for (int i = g.n; i < 4; ++i) { g.f[i] = nullptr; g.s[i] = 0; }
  hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s, g);
  ++gate_waits;
}
void CholAux::gates_disable() {
  gates_on = false; gates_broken = true;
  pipe_broken = true;
  if (bwd_pipe) { launch_pipe_fill(bwd_pipe, bwd_pipe_elems, nullptr); (void)hipDeviceSynchronize(); }   // (whatever the interrupted launch left behind)
  gate_slot.clear(); gate_slot_es.clear(); gate_seq.clear();
  if (gate_dead) (void)hipMemset(gate_dead, 0, sizeof(int));
  if (gate_dead_h) *gate_dead_h = 0;
}
void CholAux::TriCache::clear() {
  for (int* p : list) if (p) (void)hipFree(p);
  for (int* p : listC) if (p) (void)hipFree(p);
  listC.clear(); countC.clear();
  if (listA) (void)hipFree(listA);
  if (listB) (void)hipFree(listB);
  listA = listB = nullptr; countA = countB = 0;
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
  for (auto e : prof_ev2) (void)hipEventDestroy(e);
  prof_ev2.clear();
  for (auto e : panel_ev) (void)hipEventDestroy(e);
  panel_ev.clear();
  ev.clear(); prof_ev.clear();
  if (ev_zero) { (void)hipEventDestroy(ev_zero); ev_zero = nullptr; }
  if (ev_lin) { (void)hipEventDestroy(ev_lin); ev_lin = nullptr; }
  if (ev_kf) { (void)hipEventDestroy(ev_kf); ev_kf = nullptr; }
  if (ev_fill) { (void)hipEventDestroy(ev_fill); ev_fill = nullptr; }
  if (ev_xb) { (void)hipEventDestroy(ev_xb); ev_xb = nullptr; }
  if (ev_xa) { (void)hipEventDestroy(ev_xa); ev_xa = nullptr; }
  if (bwd_cnt) { (void)hipFree(bwd_cnt); bwd_cnt = nullptr; }
  if (gate_flags) { (void)hipFree(gate_flags); gate_flags = nullptr; }
  if (gate_dead) { (void)hipFree(gate_dead); gate_dead = nullptr; }
  if (gate_log) { (void)hipFree(gate_log); gate_log = nullptr; }
  if (gate_dead_h) { (void)hipHostFree(gate_dead_h); gate_dead_h = nullptr; }
  gates_on = false; gate_slot.clear(); gate_slot_es.clear(); gate_seq.clear();
  if (bwd_scr) { (void)hipFree(bwd_scr); bwd_scr = nullptr; bwd_scr_elems = 0; }
  if (bwd_pipe) { (void)hipFree(bwd_pipe); bwd_pipe = nullptr; bwd_pipe_elems = 0; }
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
  if (gate_log != nullptr && gate_log_n > 0) {   // (the caller has synchronised)
    std::vector<long long> h(2 * (size_t)gate_log_n);
    (void)hipMemcpy(h.data(), gate_log, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int i = 0; i < gate_log_n; ++i) t0 = std::min(t0, h[2 * i]);
    fprintf(stderr, "covgpu gate log [us]:");
    for (int i = 0; i < gate_log_n; ++i) {
      if (gate_log_kind[i] == 'S') fprintf(stderr, " S%d@%.1f", gate_log_tag[i], (h[2 * i] - t0) * 0.01);
      else fprintf(stderr, " G%d@%.1f+%.1f", gate_log_tag[i], (h[2 * i] - t0) * 0.01, (h[2 * i + 1] - h[2 * i]) * 0.01);
    }
    fprintf(stderr, "\n");
    gate_log_n = 0; gate_log_tag.clear(); gate_log_kind.clear();
  }
  if (!profile) return;
  for (size_t i = 0; i < prof_flops.size(); ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, prof_ev[2 * i], prof_ev[2 * i + 1]) != hipSuccess) continue;
    syrk_ms += ms; syrk_flops += prof_flops[i]; n_syrk++;
  }
  prof_flops.clear();
  for (size_t i = 0; i < prof_flops2.size(); ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, prof_ev2[2 * i], prof_ev2[2 * i + 1]) != hipSuccess) continue;
    potrf_ms += ms; potrf_flops += prof_flops2[i]; n_potrf++;
  }
  prof_flops2.clear();
}

// trailing updates given as explicit tile lists of at most this many entries (incl. the XCD padding) run as 64x64 quadrants
static const int kQuarterMax = getenv("COVGPU_QUARTER_MAX") ? atoi(getenv("COVGPU_QUARTER_MAX")) : 1024;
// look-ahead update of the rows below the next panel (stream R): tiles up to which it runs as quarter tiles
static const int kRectQuarterMax = getenv("COVGPU_RECTR_QUARTER_MAX") ? atoi(getenv("COVGPU_RECTR_QUARTER_MAX")) : 512;

void dense_cholesky_solve_raw(double* S, double* b, double* Linv, int* flag, int npad, hipStream_t st, CholAux& ax, int tstop, bool solve,
                              DenseBatch bt) {
  const int nbt = bt.n > 0 ? bt.n : 1;
  const int T = npad / kTile;
  const size_t ld = (size_t)npad;
  const size_t lds_gemm = (size_t)2 * kTile * LDT * sizeof(double);
  static std::atomic<unsigned long long> attr_seen{0};  // (per device; several host threads may arrive at once: the attribute is idempotent)
  if (first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK_TRI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK_RECT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
  }
  ax.init();
  const int NP = (T + 1) / 2;  // big panels of two tile columns
  // events per big panel: H rows-h done | B bulk done | C rows-r done | 1 potrf(t0) | 2 X(t0+1,t0) | 3 potrf(t0+1) | Rc next diagonal updated
  while ((int)ax.ev.size() < 9 * (NP + 1)) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); ax.ev.push_back(e); }
  // (a solve may factorise several matrices — arrow blocks, then the border system: launches accumulate until collect())
  if (ax.profile) while (ax.prof_ev.size() < 2 * (ax.prof_flops.size() + (size_t)NP + 1)) { hipEvent_t e; (void)hipEventCreate(&e); ax.prof_ev.push_back(e); }
  hipEvent_t* eH = ax.ev.data();
  hipEvent_t* eB = eH + (NP + 1);
  hipEvent_t* eC = eB + (NP + 1);
  hipEvent_t* e1 = eC + (NP + 1);
  hipEvent_t* e2 = e1 + (NP + 1);
  hipEvent_t* e3 = e2 + (NP + 1);
  hipEvent_t* eRc = e3 + (NP + 1);
  hipEvent_t* eHp = eRc + (NP + 1);  // rows h updated with column t0 (their last TRSM then runs on the chain's own stream)
  hipEvent_t* eA = eHp + (NP + 1);   // bulk(P) has updated the NEXT-BUT-ONE panel's two tile columns (the first of its two launches; == eB where the bulk is one launch)

  // C tiles (rows [r0, r1), tile columns [tc0, tc0+ntc)) -= A[rows, K] A[tc.., K]^T, K = tiles kt0.. (KD columns); lower part only
  auto rect = [&](int r0, int r1, int tc0, int ntc, int kt0, int KD, hipStream_t s2, bool quad) {
    if (r1 <= r0 || ntc <= 0 || KD <= 0) return;
    GemmArgs g{S, ld, kt0 * kTile, KD, r0 * kTile, tc0 * kTile, tc0 * kTile, r1 - r0, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab, bt.own_dims};
    if (quad) hipLaunchKernelGGL((k_gemm_abt_q<MODE_SYRK_RECT, 64, 64>), dim3(ntc, r1 - r0, 4 * nbt), dim3(256), (size_t)(64 + 64) * (KCQ + kLdsPad) * sizeof(double), s2, g);
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
  // ordering between the four streams: device flags (CholAux::record / wait) or, with COVGPU_GATES=0, HIP events. Waits of one stream that stand
  // side by side are ONE gate launch.
  auto wait = [&ax](hipStream_t s2, hipEvent_t e0, hipEvent_t e1 = nullptr, hipEvent_t e2 = nullptr, hipEvent_t e3 = nullptr) { ax.wait(s2, e0, e1, e2, e3); };
  auto record = [&ax](hipEvent_t e, hipStream_t s2, int tag = 0) { ax.record(e, s2, tag); };
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
  // K range front `a` really runs in panel Pp's rank update (GemmArgs::own): for the flop count of the profiled run
  auto kd_front = [&](int Pp, int a, int chunk) {
    if (bt.own_dims_h == nullptr) return kd(Pp);
    const int real = bt.own_dims_h[a] - 2 * Pp * kTile;
    return real <= 0 ? 0 : std::min(kd(Pp), ((real + chunk - 1) / chunk) * chunk);
  };
  if (bt.tri_slot >= (int)ax.tri_lev.size()) ax.tri_lev.resize(bt.tri_slot + 1);
  CholAux::TriCache& tc = bt.tri_slot >= 0 ? ax.tri_lev[bt.tri_slot] : ax.tri0;
  // live tiles (i, j), j <= i, of the triangle that starts at tile tb, for the fronts whose interior reaches panel column t0;
  // part 0: all | 1: rows i < split_ta | 2: rows i >= split_ta | 3: tile columns j < 2 and rows i < split_ta | 4: the others. XCD-balanced, interleaved (position p runs on XCD p % 8).
  auto build_list = [&](int t0, int tb, int part, int*& d_out, int& n_out) {
    d_out = nullptr; n_out = 0;
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
              if (live(tb + i) && live(tb + j) && (part == 0 || (part <= 2 && (i < bt.split_ta) == (part == 1)) || (part >= 3 && (j < 2 || i < bt.split_ta) == (part == 3)))) grp.push_back((a << 20) | (i << 10) | j);
          if (grp.empty()) continue;
          int best = 0;
          for (int x = 1; x < 8; ++x) if (q[x].size() < q[best].size()) best = x;
          q[best].insert(q[best].end(), grp.begin(), grp.end());
        }
    }
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) longest = std::max(longest, q[x].size());
    if (longest == 0) return true;
    std::vector<int> lst(8 * longest, -1);
    for (int x = 0; x < 8; ++x) for (size_t k = 0; k < q[x].size(); ++k) lst[8 * k + x] = q[x][k];
    if (hipMalloc((void**)&d_out, lst.size() * sizeof(int)) != hipSuccess) { d_out = nullptr; return false; }
    (void)hipMemcpy(d_out, lst.data(), lst.size() * sizeof(int), hipMemcpyHostToDevice);
    n_out = (int)lst.size();
    return true;
  };
  // Two launches per bulk update (round 5): first the tiles of the next-but-one panel's two tile columns, then the rest. Everything on or beside the
  // chain that follows a bulk update — the next diagonal look-ahead, the look-ahead of the next panel's rows — writes those two tile columns only and
  // waits for the FIRST launch (eA). With one launch the chain of a big front waited for the whole previous bulk update every panel: on the 5-agent
  // map's root (3 726 unknowns, 15 panels, bulk 100 us) the period was bulk -> diagonal look-ahead -> bulk = 165 us instead of the chain's 128.
  static const bool col_split = getenv("COVGPU_BULK_SPLIT") == nullptr || atoi(getenv("COVGPU_BULK_SPLIT")) != 0;
  const int key = ((T * 4096 + nbt) * 8 + bt.split_ta) * 2 + (col_split ? 1 : 0);
  if (bt.live_h != nullptr && tc.key != key) {  // live-tile lists of every panel's bulk update (static per problem)
    tc.clear();
    tc.key = key;
    tc.list.assign(NP, nullptr); tc.count.assign(NP, 0); tc.listC.assign(NP, nullptr); tc.countC.assign(NP, 0);
    for (int P = 0; P < NP && P < Pstop; ++P) {
      // (the LAST panel of a partial factorisation applies its whole trailing update in one launch: triangle from t0 + 2)
      const bool last = Pstop < NP && P == Pstop - 1;
      const int t0 = 2 * P, tb = last ? t0 + 2 : t0 + 4;
      if (tb >= T) break;
      bool ok = true;
      if (last && bt.split_ta > 0) ok = build_list(t0, tb, 1, tc.listA, tc.countA) && build_list(t0, tb, 2, tc.listB, tc.countB);
      else if (col_split && !last) ok = build_list(t0, tb, 3, tc.listC[P], tc.countC[P]) && build_list(t0, tb, 4, tc.list[P], tc.count[P]);   // (the last panel's update is ONE launch on the chain's stream)
      else ok = build_list(t0, tb, 0, tc.list[P], tc.count[P]);
      if (!ok) { tc.clear(); break; }
    }
  }
  // k_potrf_panel of big panel P, all fronts of the batch; profiling: an event pair around the launch and its algorithmic flops
  // (per front n^3 / 3 for the factorisation + n^2 for the forward substitution riding along, n = the front's REAL columns in the panel)
  // records of the chain's stream waiting to be published by the next panel factorisation's first thread (CholAux::publish_handle)
  DevSignal pend[2]; hipEvent_t pend_ev[2] = {nullptr, nullptr}; int pend_tag[2] = {0, 0}; int npend = 0;
  auto flush_pending = [&]() {   // no factorisation follows (or none was launched): an ordinary launch publishes them
    if (npend == 0) return;
    hipLaunchKernelGGL(k_signal2, dim3(1), dim3(64), 0, M, pend[0], npend > 1 ? pend[1] : DevSignal());
    npend = 0;
  };
  auto potrf = [&](int t0, int w, int nbp) {
    const bool prof = ax.profile && nbp != 0;
    if (prof) {
      while (ax.prof_ev2.size() < 2 * (ax.prof_flops2.size() + 1)) { hipEvent_t e; (void)hipEventCreate(&e); ax.prof_ev2.push_back(e); }
      (void)hipEventRecord(ax.prof_ev2[2 * ax.prof_flops2.size()], M);
    }
    const DevSignal sa = npend > 0 ? pend[0] : DevSignal(), sb = npend > 1 ? pend[1] : DevSignal();
    bool launched;
    if (bt.plist != nullptr) launched = launch_potrf_panel(S, ld, t0, w, Linv, flag, b, npad, nbt, bt.sM, bt.sL, bt.sR, M, bt.tab, nbp, bt.own_dims, bt.plist, bt.pbig_h[t0 / 2], bt.psmall_h[t0 / 2], sa, sb);
    else launched = launch_potrf_panel(S, ld, t0, w, Linv, flag, b, npad, nbt, bt.sM, bt.sL, bt.sR, M, bt.tab, nbp, nullptr, nullptr, 0, 0, sa, sb);
    if (launched) npend = 0; else flush_pending();
    if (prof) {
      double fl = 0.0;
      for (int a = 0; a < nbt; ++a) {
        const double n = bt.own_dims_h != nullptr ? (double)std::max(0, std::min(w * kTile, bt.own_dims_h[a] - t0 * kTile)) : (double)(w * kTile);
        fl += n * n * n / 3.0 + n * n;
      }
      (void)hipEventRecord(ax.prof_ev2[2 * ax.prof_flops2.size() + 1], M);
      ax.prof_flops2.push_back(fl);
    }
  };
  int Plast = NP - 1;
  bool split_last = false;  // the last panel's bulk update was left running on B for the caller (DenseBatch::split_ta)
  bool tail_on_chain = false;  // the last panel ran whole on the chain's own stream: nothing of it to join (every event packet on
                               // the chain's stream is a few microseconds between two dependent kernels)
  static const bool early_wait2_env = getenv("COVGPU_EARLY_WAIT") == nullptr || atoi(getenv("COVGPU_EARLY_WAIT")) != 0;
  static const bool trace2 = getenv("COVGPU_TRACE_PANELS") != nullptr && atoi(getenv("COVGPU_TRACE_PANELS")) >= 2;   // dev aid: per-kernel marks, tag 100 (P + 1) + k
  for (int P = 0; P < NP; ++P) {
    const int t0 = 2 * P, w = (T - t0 >= 2) ? 2 : 1;
    const int h0 = (t0 + 2 < T) ? t0 + 2 : T, h1 = (t0 + 4 < T) ? t0 + 4 : T;  // rows h = [h0, h1), rows r = [h1, T)
    // ---- look-ahead part of SYRK(P-1) (K = the 256 columns of panel P-1) on this panel's two tile columns; the
    //      2x2 diagonal part was enqueued on M at the end of the previous iteration
    if (P > 0) {
      if (h1 > h0) {
        // L rows h0.. were rest rows of panel P-1 (eC) | B operand: L rows t0, t0+1 — their last TRSM ran on the chain's stream (eH) |
        // bulk(P-2) was the previous writer of these tiles (its first launch: these two tile columns, eA)
        wait(H, eC[P - 1], eH[P - 1], P >= 2 ? eA[P - 2] : nullptr);
        rect(h0, h1, t0, w, t0 - 2, kd(P - 1), H, true);
        if (trace2) ax.mark(H, 100 * (P + 1) + 6);   // rows h carry panel P-1
        record(eHp[P], H, 100 * (P + 1) + 17);
      }
      if (T > h1) {
        wait(R, eH[P - 1], P >= 2 ? eA[P - 2] : nullptr);   // B operand: L rows t0, t0+1 (rows h of panel P-1) | previous writer
        // (round 5: as quarter tiles while the launch is small — a full tile is one workgroup's 16-chunk K loop, 50-57 us of latency that the
        //  rest rows' substitution and, behind it, the bulk update and the last panel's substitution wait for)
        rect(h1, T, t0, w, t0 - 2, kd(P - 1), R, (T - h1) * w * nbt <= kRectQuarterMax);
        if (trace2) ax.mark(R, 100 * (P + 1) + 7);   // rows r carry panel P-1
        record(e2[P], R, 100 * (P + 1) + 14);  // rows r carry panel P-1's update
      }
    }
    if (P == Pstop) {  // only the look-ahead updates of the last eliminated panel; nothing of this panel is factored
      flush_pending();
      record(eH[P], H, 100 * (P + 1) + 10);
      record(eC[P], R, 100 * (P + 1) + 12);
      record(eB[P], B, 100 * (P + 1) + 11);
      record(eA[P], B, 100 * (P + 1) + 18);
      Plast = P;
      break;
    }
    // ---- M: critical chain
    ax.mark(M, P);
    if (Pstop < NP && P == Pstop - 1) {
      // LAST panel of a partial factorisation (every multifrontal front, every arrow block): nothing is factored after it, so
      // the look-ahead split of its trailing update (next panel's rows / diagonal / rest rows / bulk: four launches on three
      // streams, ~100 us of event hops per front level) buys nothing — factor, solve ALL rows below, update the WHOLE trailing
      // triangle, three dependent launches on the chain's own stream.
      const int nbp = bt.own_max > 0 ? std::max(0, std::min(8 * w, (bt.own_max - t0 * kTile + 15) / 16)) : -1;
      potrf(t0, w, nbp);
      if (trace2) ax.mark(M, 100 * (P + 1) + 1);   // potrf done
      if (P == 0 && bt.pre_trsm != nullptr) wait(M, bt.pre_trsm);   // rows below the first panel: second half of the caller's extend-add
      bool split_done = false;
      if (T > h0) {
        const bool early_wait2 = early_wait2_env;
        const bool split = bt.split_ta > 0 && bt.live_h != nullptr && kd(P) > 0 && (tc.listA != nullptr || tc.listB != nullptr);
        const bool waitedA = early_wait2 && split && P >= 1;
        // rows h / rest rows carry panel P-1 | (beside them instead of between the substitution and the update: see the multi-panel branch) bulk(P-1)'s first launch
        if (P > 0) wait(M, h1 > h0 ? eHp[P] : nullptr, T > h1 ? e2[P] : nullptr, waitedA ? eA[P - 1] : nullptr);
        // (measured and dropped: solving only the rows the parents' first panel receives here and the others on the bulk stream —
        //  the bulk stream's leg (rest rows, rest of the update, second half of the extend-add) is what the next level's
        //  substitutions wait for, and it only got longer: 3.37 vs 3.33 ms)
        launch_trsm_sub(S, ld, t0, w, h0, T, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, M, true, bt.tab, nbp, bt.own_dims);
        if (split) { record(eH[P], M, 100 * (P + 1) + 10); wait(B, eH[P]); }
        // bulk(P-1) was the previous writer of the trailing tiles. The part of this update that stays on the chain's stream (rows < split_ta: what
        // build_list puts into the first launch beside tile columns 0, 1) only meets the FIRST launch of that bulk update — on the 5-agent map's upper levels (borders of
        // 2 000 unknowns) the whole of it is 160 us, and the next level's first panel waited for it; the rest follows it on the bulk stream anyway
        if (P >= 1 && !waitedA) wait(M, split ? eA[P - 1] : eB[P - 1]);
        const int tb = h0, nt = T - tb;
        auto syrk = [&](hipStream_t s2, const int* list, int count, double flops) {
          GemmArgs g{S, ld, t0 * kTile, kd(P), tb * kTile, tb * kTile, tb * kTile, nt, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab, bt.own_dims};
          g.tri = list; g.beta0_off = bt.beta0_off; g.beta0 = bt.beta0;
          if (ax.profile) (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size()], s2);
          // a short list runs at the LATENCY of one workgroup's K loop (16 chunks of 64 MFMAs per wave): as 64x64 quadrants it is
          // four times as many workgroups, each four times shorter
          if (list != nullptr && count <= kQuarterMax)
            hipLaunchKernelGGL((k_gemm_abt_q<MODE_SYRK_TRI, 64, 64>), dim3(count, 1, 4), dim3(256), (size_t)(64 + 64) * (KCQ + kLdsPad) * sizeof(double), s2, g);
          else if (list != nullptr) hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(count, 1), dim3(256), lds_gemm, s2, g);
          else {
            const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
            hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(nblk, nbt), dim3(256), lds_gemm, s2, g);
          }
          if (ax.profile) {
            (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size() + 1], s2);
            ax.prof_flops.push_back(flops);
          }
        };
        if (kd(P) > 0) {
          double pairs = 0.0, pairsA = 0.0;   // flops of the whole update | of its first split_ta tile rows
          const double per_k = 2.0 * kTile * kTile;
          for (int a = 0; a < nbt; ++a) {
            if (bt.live_h == nullptr) { pairs += (double)nt * (nt + 1) / 2 * per_k * kd(P); continue; }
            const int nI = bt.live_h[2 * a], nO = bt.live_h[2 * a + 1];
            if (t0 >= nI) continue;
            int nl = 0;
            for (int t = tb; t < T; ++t) nl += (t < nI || (t >= bt.tI && t - bt.tI < nO)) ? 1 : 0;
            const double kf = kd_front(P, a, KCQ);
            pairs += (double)nl * (nl + 1) / 2 * per_k * kf;
            const int na = std::min(nl, bt.split_ta);
            pairsA += (double)na * (na + 1) / 2 * per_k * kf;
          }
          if (split) {
            // look-ahead across levels: the tiles the parents' first panel receives on the chain's stream, the rest on the bulk stream
            if (tc.countA > 0) syrk(M, tc.listA, tc.countA, pairsA);
            if (tc.countB > 0) syrk(B, tc.listB, tc.countB, pairs - pairsA);
            record(eB[P], B, 100 * (P + 1) + 11);
            split_done = true;
          } else {
            const bool listed = bt.live_h != nullptr && P < (int)tc.list.size() && tc.list[P] != nullptr;
            if (!listed || tc.count[P] > 0) syrk(M, listed ? tc.list[P] : nullptr, listed ? tc.count[P] : 0, pairs);
          }
        }
      }
      Plast = P; split_last = split_done; tail_on_chain = true;
      break;
    }
    bool bulk_wait_rc = false;
    {
      // 256-column chain (k_panel.hip): one workgroup factors the whole diagonal block, rows h follow on the same stream by
      // block substitution, rows r on theirs — three dependent launches per panel instead of six
      const int nbp = bt.own_max > 0 ? std::max(0, std::min(8 * w, (bt.own_max - t0 * kTile + 15) / 16)) : -1;
      const double bulk_pairs = (T - (t0 + 4) > 0) ? 0.5 * (double)(T - (t0 + 4)) * (T - (t0 + 4) + 1) * nbt : 0.0;
      // (round 4: the threshold was 700 tile pairs — every panel of the 5-agent map's fronts. With the panel factorisation at 60 us the
      //  period of a multi-panel front was no longer the chain but the cycle  look-ahead(P) -> rest rows(P) -> bulk(P) -> look-ahead(P+1),
      //  all of it behind the ONE event of the chain-bound form: 140-160 us per panel. Events after every kernel: 284 -> 298 it/s, and
      //  no loss on the one-panel fronts. COVGPU_CHAIN_PAIRS restores a threshold.)
      static const double chain_pairs = getenv("COVGPU_CHAIN_PAIRS") ? atof(getenv("COVGPU_CHAIN_PAIRS")) : 0.0;
      static const bool early_wait = getenv("COVGPU_EARLY_WAIT") == nullptr || atoi(getenv("COVGPU_EARLY_WAIT")) != 0;
      static const bool merge_trsm = getenv("COVGPU_TRSM_MERGE") != nullptr && atoi(getenv("COVGPU_TRSM_MERGE")) != 0;
      const bool chain_bound = bulk_pairs <= chain_pairs;
      potrf(t0, w, nbp);
      // rows below the first panel: second half of the caller's extend-add. IN FRONT of the record below — the rest rows' substitution on stream R
      // starts from that record and inherits this wait
      if (P == 0 && bt.pre_trsm != nullptr) wait(M, bt.pre_trsm);
      // Every event packet on this stream sits between two dependent kernels of the chain, a few microseconds each. While the bulk
      // update is short (the period of the factorisation is the chain: small fronts, the tail of big ones) there is ONE event per
      // panel, eH, recorded after the look-ahead update of the next diagonal block — the rest rows (which only need the factored
      // panel), the next panel's rows on stream H and the bulk update all start from it, two small kernels later than they could.
      // While the bulk update is long (the period is the bulk) they start as early as possible: an event after each kernel.
      // merge_trsm (round 5, opt-in: COVGPU_TRSM_MERGE=1): rows h AND the rest rows in ONE substitution launch on the chain's stream — every 16-row
      // slab is a workgroup of its own, so the launch is as long as one slab while workgroup slots are free — instead of the rest rows on stream R
      // behind an event: a record whose waiter is blocked on it at that moment costs the RECORDING stream ~13 us as well
      // (profiles/r05z_iteration_timeline.csv: both substitutions start 14 us after the factorisation ends), a record nobody is waiting for yet ~6.
      // Measured: 233.5 / 233.1 against 232.6 / 232.0 it/s, configs[4] 20.0 against 20.3 — within noise, so the default stays the round-4 form.
      const bool merge = merge_trsm && !chain_bound && h1 > h0 && T > h1;
      bool waitedA = false;
      hipEvent_t rec1 = (!chain_bound && !merge) ? e1[P] : nullptr;   // "panel P factored": the rest rows' substitution (stream R) starts from it
      if (h1 > h0) {
        // rows h carry the look-ahead update of panel P-1 (stream H, above) | ... and, merged, the rest rows theirs (stream R) | bulk(P-1)'s first launch.
        // (round 6: the record behind the factorisation and these waits are ONE launch — CholAux::sync)
        waitedA = early_wait && P >= 1 && P + 1 < NP;
        ax.sync(M, rec1, 100 * (P + 1) + 13, nullptr, 0, P > 0 ? eHp[P] : nullptr, merge && P > 0 ? e2[P] : nullptr, waitedA ? eA[P - 1] : nullptr);
        rec1 = nullptr;
        // (round 5: the wait of the next-diagonal update below for bulk(P-1)'s first launch — 30-40 us of slack — rides along with the one above.
        //  Measured: no difference (231.1 / 230.5 against 231.4 / 231.2 it/s) — a wait whose event completed long ago costs nothing; the 13-14 us
        //  between two kernels of this stream come from the RECORD behind the first when its waiter is blocked on it at that moment: DESIGN.md 4.6)
        launch_trsm_sub(S, ld, t0, w, h0, merge ? T : h1, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, M, true, bt.tab, nbp, bt.own_dims);
        if (trace2) ax.mark(M, 100 * (P + 1) + 2);   // rows h solved
      }
      if (rec1 != nullptr) record(rec1, M, 100 * (P + 1) + 13);
      // (round 6: "rows h solved" is published together with "next diagonal block updated", one launch behind that small update instead of one
      //  launch in front of it and one behind: its waiters — the next panel's look-ahead on stream H — have ~50 us of slack)
      static const bool late_eh = getenv("COVGPU_LATE_EH") == nullptr || atoi(getenv("COVGPU_LATE_EH")) != 0;
      const bool eh_now = !chain_bound && !(late_eh && P + 1 < NP);
      if (eh_now) record(eH[P], M, 100 * (P + 1) + 10);
      // ---- M: look-ahead part of SYRK(P) on the next panel's 2x2 diagonal tiles
      if (P + 1 < NP) {
        const int u0 = t0 + 2, uw = (T - u0 >= 2) ? 2 : 1;
        if (P >= 1 && !waitedA) wait(M, eA[P - 1]);          // bulk(P-1) was the previous writer of these tiles (its first launch)
        rect(u0, u0 + uw, u0, uw, t0, kd(P), M, true);
        if (trace2) ax.mark(M, 100 * (P + 1) + 3);   // next diagonal block updated
      }
      // (A bulk launched at the same instant as the next diagonal update takes every workgroup slot first and the chain waits
      //  ~75 us for the first round of tiles to retire: the bulk starts after that small kernel in either regime.)
      if (chain_bound || eh_now) record(chain_bound ? eH[P] : eRc[P], M, 100 * (P + 1) + (chain_bound ? 10 : 16));
      else {
        // (round 6, OPT-IN: COVGPU_RECORD_RIDE=1) both records ride with the NEXT panel's factorisation, whose first thread publishes them — one launch
        // less per panel (-1 % of the factorisation). Their waiters (stream H's look-ahead of the next panel, the bulk update) are then enqueued BEFORE
        // the launch that publishes: harmless while every stream has a hardware queue of its own, a deadlock (until the gate's timeout) as soon as
        // two streams of the process share one — a gate in front of the publishing launch in the same queue holds it back. Several contexts in one
        // process (virtual ranks, the facade's context beside another) do share queues: tests/test_gpu_shard.py hung on it. Default: off —
        // every gate is enqueued after the launch that publishes its record, and the ordering cannot deadlock whatever the queue mapping.
        static const bool ride = getenv("COVGPU_RECORD_RIDE") != nullptr && atoi(getenv("COVGPU_RECORD_RIDE")) != 0;
        if (ride && ax.gates_on && npend == 0) {
          pend[0] = ax.publish_handle(eH[P], M, 100 * (P + 1) + 10); pend[1] = ax.publish_handle(eRc[P], M, 100 * (P + 1) + 16);
          npend = (pend[0].flag != nullptr && pend[1].flag != nullptr) ? 2 : 0;
          if (npend == 0) { record(eH[P], M, 100 * (P + 1) + 10); record(eRc[P], M, 100 * (P + 1) + 16); }   // (out of slots: HIP events took over inside publish_handle's fallback)
        } else ax.sync(M, eH[P], 100 * (P + 1) + 10, eRc[P], 100 * (P + 1) + 16);
      }
      if (T > h1 && !merge) {
        wait(R, chain_bound ? eH[P] : e1[P]);   // (measured, round 4: waiting for rows h instead — so that the chain's substitution runs alone — 299.9 -> 297.5 it/s)
        launch_trsm_sub(S, ld, t0, w, h1, T, Linv, b, npad, nbt, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, R, false, bt.tab, nbp, bt.own_dims);
        if (trace2) ax.mark(R, 100 * (P + 1) + 4);   // rest rows solved
      }
      if (merge) wait(R, eH[P]);   // (eC then stands for "every row below panel P is solved" as before: its waiters — the bulk update, stream H — need not change)
      record(eC[P], R, 100 * (P + 1) + 12);
      bulk_wait_rc = !chain_bound;   // (measured again in round 4 with the 60 us panel: without this wait 298.6 -> 292.7 it/s)
    }
    // ---- B: bulk of SYRK(P), triangle starting two tile columns further (those belong to the look-ahead)
    const int tb = t0 + 4, nt = T - tb;
    bool recA = false;
    wait(B, eC[P], bulk_wait_rc ? eRc[P] : nullptr);
    if (nt > 0 && kd(P) > 0) {
      const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
      GemmArgs g{S, ld, t0 * kTile, kd(P), tb * kTile, tb * kTile, tb * kTile, nt, nullptr, nullptr, nullptr, bt.sM, bt.sL, bt.sR, bt.live, bt.tI, nullptr, bt.tab, bt.own_dims};
      // arrow buffers: the live tiles of this panel's update as an explicit, XCD-balanced list (built once per problem).
      // The implicit triangle grid x batch launched ~2.6k workgroups of which ~400 did work, with every batch's first
      // supertile on XCD 0: 17 TFLOP/s.
      const bool listed = bt.live_h != nullptr && P < (int)tc.list.size() && tc.list[P] != nullptr;
      if (listed) g.tri = tc.list[P];
      g.beta0_off = bt.beta0_off; g.beta0 = bt.beta0;
      double flops = 0.0;  // of the tile pairs that do work, over the K range each front really runs
      for (int a = 0; a < nbt; ++a) {
        if (bt.live_h == nullptr) { flops += (double)nt * (nt + 1) / 2 * 2.0 * kTile * kTile * kd(P); continue; }
        const int nI = bt.live_h[2 * a], nO = bt.live_h[2 * a + 1];
        if (t0 >= nI) continue;
        int nl = 0;
        for (int t = tb; t < T; ++t) nl += (t < nI || (t >= bt.tI && t - bt.tI < nO)) ? 1 : 0;
        flops += (double)nl * (nl + 1) / 2 * 2.0 * kTile * kTile * kd_front(P, a, KC);
      }
      const int cntC = (listed && P < (int)tc.listC.size() && tc.listC[P] != nullptr) ? tc.countC[P] : 0;
      if (!listed || tc.count[P] > 0 || cntC > 0) {
        if (ax.profile) (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size()], B);
        // A long bulk update keeps every CU full for hundreds of microseconds, and the next panel's factorisation — one 16-wave workgroup that needs an
        // EMPTY CU — waits for its tail (12-agent map, root: period = bulk + the whole exposed chain). In pieces of kBulkChunk tiles the chip drains
        // between two launches and the waiting workgroup (higher stream priority) gets its CU. (0: one launch.)
        static const int kBulkChunk = getenv("COVGPU_BULK_CHUNK") ? atoi(getenv("COVGPU_BULK_CHUNK")) : 0;
        auto tri = [&](const int* list, int count) {
          if (count <= kQuarterMax) { g.tri = list; hipLaunchKernelGGL((k_gemm_abt_q<MODE_SYRK_TRI, 64, 64>), dim3(count, 1, 4), dim3(256), (size_t)(64 + 64) * (KCQ + kLdsPad) * sizeof(double), B, g); return; }
          const int step = kBulkChunk > 0 ? std::max(512, (kBulkChunk / 8) * 8) : count;
          for (int o = 0; o < count; o += step) {
            g.tri = list + o;
            hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(std::min(step, count - o), 1), dim3(256), lds_gemm, B, g);
          }
        };
        if (cntC > 0) {   // the next-but-one panel's two tile columns first: what the chain waits for
          tri(tc.listC[P], cntC);
          record(eA[P], B, 100 * (P + 1) + 18); recA = true;
          if (trace2) ax.mark(B, 100 * (P + 1) + 8);   // first bulk launch done
        }
        if (listed) { if (tc.count[P] > 0) tri(tc.list[P], tc.count[P]); }
        else hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(nblk, nbt), dim3(256), lds_gemm, B, g);
        if (ax.profile) {
          (void)hipEventRecord(ax.prof_ev[2 * ax.prof_flops.size() + 1], B);
          ax.prof_flops.push_back(flops);
        }
      }
    }
    if (trace2) ax.mark(B, 100 * (P + 1) + 5);   // bulk update done
    record(eB[P], B, 100 * (P + 1) + 11);
    if (!recA) record(eA[P], B, 100 * (P + 1) + 18);
  }
  flush_pending();
  wait(M, !split_last && !tail_on_chain ? eB[Plast] : nullptr, Plast >= 1 ? eB[Plast - 1] : nullptr, !tail_on_chain ? eC[Plast] : nullptr, !tail_on_chain ? eH[Plast] : nullptr);
  if (!solve) return;
  // y = L^-1 b was formed along the way (potrf: y_p = L_pp^-1 b_p; every TRSM: b[rows] -= L[rows, p] y_p) and lives
  // in b[npad .. 2 npad). Remaining: L^T x = y.
  dense_backward_solve(S, b, Linv, npad, st, T, T, bt);
}

// Fronts of at most this many interior tiles run their whole backward substitution in ONE launch (k_bwd_front: the last workgroup solves the
// interior tiles one after the other). Measured in round 5 on the corrected 5-agent map, whose upper levels hold fronts of 5-6 tiles: the
// serial part costs ~15 us per tile (dependent loads of one workgroup), a launch per tile 8.6 — 4: 212.8 it/s, 8: 211.8, 16: 200.0.
int bwd_front_max_tiles() {
  static const int v = getenv("COVGPU_BWD_FRONT_TILES") ? std::max(1, atoi(getenv("COVGPU_BWD_FRONT_TILES"))) : 4;
  return v;
}
// Fronts of at least this many interior tiles: one launch, one workgroup per tile, hand-overs inside the launch (k_panel.hip: k_bwd_pipe).
// COVGPU_BWD_PIPE=0 disables it.
int bwd_pipe_min_tiles() {
  static const int v = (getenv("COVGPU_BWD_PIPE") && atoi(getenv("COVGPU_BWD_PIPE")) == 0) ? (1 << 30) : getenv("COVGPU_BWD_PIPE_MIN") ? std::max(1, atoi(getenv("COVGPU_BWD_PIPE_MIN"))) : 2;
  return v;
}
// L^T x = y for the factored tile columns [0, tfact) of an npad-order matrix; for tile rows p in [tfact, tend) x_p is
// GIVEN (already in b[p*128 ..]) and only its contribution L[rows p, cols < tfact*128]^T x_p is taken out of y.
void dense_backward_solve(double* S, double* b, double* Linv, int npad, hipStream_t st, int tfact, int tend, DenseBatch bt) {
  const int nbt = bt.n > 0 ? bt.n : 1;
  const size_t ld = (size_t)npad;
  {
    // multifrontal front of few interior tiles: given rows and every interior tile in ONE launch (k_panel.hip: k_bwd_front)
    const int nt_real = bt.own_max > 0 ? std::min(tfact, (bt.own_max + kTile - 1) / kTile) : tfact;
    if (bt.bwd_pipe != nullptr && bt.pipe_dead != nullptr && bt.xfer.gidx != nullptr && bt.tab != nullptr && bt.live != nullptr && nt_real >= bwd_pipe_min_tiles() && nbt <= 65535) {
      launch_bwd_pipe(S, tfact, nt_real, ((tend - tfact) * kTile + kPipeChunk - 1) / kPipeChunk, b + npad, Linv, nbt, bt.sL, bt.sR, st, bt.tab, bt.live, bt.xfer, bt.bwd_pipe, bt.pipe_dead,
                      bt.pipe_dead_h, bt.pipe_timeout_s);
      return;
    }
    if (bt.bwd_cnt != nullptr && bt.bwd_scr != nullptr && bt.xfer.gidx != nullptr && bt.tab != nullptr && bt.live != nullptr && nt_real >= 1 && nt_real <= bwd_front_max_tiles() &&
        nbt <= 65536) {
      const int nchunk = std::max(1, ((tend - tfact) * kTile + 255) / 256);
      launch_bwd_front(S, tfact, nt_real, nchunk, b + npad, Linv, nbt, bt.sL, bt.sR, st, bt.tab, bt.live, bt.xfer, bt.bwd_cnt, bt.bwd_scr);
      return;
    }
  }
  if (tend > tfact) {  // all given rows in one launch (k_panel.hip)
    launch_bwd_given(S, ld, tfact * kTile, tend * kTile, b + npad, b, tfact * kTile, nbt, bt.sM, bt.sR, st, bt.tab, bt.live, bt.tI, bt.xfer);
    tend = tfact;
  }
  if (bt.own_max > 0 && tend <= tfact) tend = std::min(tend, (bt.own_max + kTile - 1) / kTile);  // all-padding interior tiles: x = 0
  for (int p = tend - 1; p >= 0; --p) {
    const bool given = p >= tfact;
    const int ncol = given ? tfact * kTile : p * kTile;
    const int nb = (ncol + 31) / 32;
    if (given && nb == 0) continue;
      launch_bwd_step_sub(S, ld, p, given ? nullptr : Linv + (size_t)p * kTile * kTile, b + npad, b, ncol, nb > 0 ? nb : 1, nbt, bt.sM, bt.sL, bt.sR, st, bt.tab,
                          bt.live, bt.tI, p == 0 ? bt.xfer : BwdXfer());
  }
}

}  // namespace covgpu
