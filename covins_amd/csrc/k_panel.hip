// k_panel.hip — the serial chain of the dense FP64 Cholesky (k_chol.hip), 256 columns per step.
//
// Replaces, like k_chol.hip, the linear-algebra half of ceres::Solve(SPARSE_SCHUR) (optimization_be.cpp:560-567,
// 1024-1031). With the fronts of the elimination tree the factorisation of the 5-agent map is 15 serial
// panels of pure latency. In round 2a a 256-column panel was potrf(128) -> TRSM -> rank-128 update -> potrf(128) -> TRSM ->
// next-diagonal update, six dependent launches of 20-80 us each with a 128x128 explicit inverse in the middle. Here the
// same panel is three launches and no large inverse:
//   k_potrf_panel   ONE workgroup (16 waves) factors the whole 256x256 diagonal block. The trailing 16x16 tiles live in
//                   REGISTERS in the MFMA accumulator layout for the whole kernel (119 tiles over the twelve waves of SIMDs
//                   1-3); only the current 16-column panel travels through LDS (double-buffered). Two barriers per block
//                   column: wave 0 carries the serial chain alone (diagonal block + its inverse in one sweep IN REGISTERS,
//                   rows exchanged by DPP row broadcasts -> the tile below it -> the next diagonal tile), the tile waves solve
//                   the rest of the panel against the block inverse on the matrix core and apply the panel to every trailing
//                   tile; L / y leave by whoever has the time. The right-hand side rides along as one more row. Leaves L in
//                   place, y = L^-1 b, and the INVERSES OF THE SIXTEEN 16x16 DIAGONAL BLOCKS. (Round 3: 8 waves, the sweep
//                   through LDS round trips, 87 us per launch; round 4: 60 us. tools/panel_probe.hip, tools/valu_probe.hip.)
//   k_trsm_sub4     X = A L^-T for 16-row slabs below the panel, FOUR waves per slab (one per SIMD): block forward
//                   substitution on Z = X^T kept in accumulator layout — a finished 16x16 block Z_j IS the B operand of the
//                   trailing updates acc_i -= L_ij Z_j (the C/D layout of v_mfma_f64_16x16x4 equals its B layout), and
//                   Z_j = Dinv_j acc_j needs only the small block inverses. Exact substitution between blocks: better
//                   conditioned than the product with a 128x128 explicit inverse it replaces. (k_trsm_sub: the same with one
//                   wave per slab, bound by one SIMD's matrix pipe; COVGPU_TRSM4=0.)
//   k_bwd_step_sub  backward substitution per 128-tile with the same block inverses.
// A logical row permutation makes every MFMA operand a contiguous 32-byte load: hardware k-slot (lane>>4, step s) carries
// logical index 4*(lane>>4)+s instead of (lane>>4)+4*s — consistently for A and B, so products are unchanged.
#include <cstdio>
#include <cstdlib>

#include <algorithm>

#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {

typedef double v4f64 __attribute__((ext_vector_type(4)));

#ifdef COVGPU_PROBE
__device__ long long g_pprobe[8];
__device__ long long g_pstep[4][16];   // per step: wave 0 sweep done, wave 0 past X, wave 0 past Y, probed tile wave's updates done (relative to kernel start)
#define PSTEP(k, j, t0) do { if ((threadIdx.x & 63) == 0) g_pstep[k][j] = clock64() - (t0); } while (0)
__device__ long long g_parr[16][16];   // arrival of every wave at barrier X(j)
#define PARR(j, t0) do { if ((threadIdx.x & 63) == 0) g_parr[threadIdx.x >> 6][j] = clock64() - (t0); } while (0)
#ifndef PPROBE_WAVE
#define PPROBE_WAVE 1
#endif
// (accumulated in registers and written once: a global read-modify-write per phase costs more than the phases themselves)
#define PPROBE_DECL() long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PPROBE_ACC(i, t0) do { pacc[i] += clock64() - (t0); } while (0)
#define PPROBE_T0() clock64()
#define PPROBE_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 1; i_ < 6; ++i_) g_pprobe[i_] = pacc[i_]; if (threadIdx.x == 64 * PPROBE_WAVE) { g_pprobe[6] = pacc[6]; g_pprobe[7] = pacc[7]; } } while (0)
#else
#define PPROBE_DECL() do {} while (0)
#define PPROBE_ACC(i, t0) do {} while (0)
#define PPROBE_T0() 0
#define PPROBE_FLUSH() do {} while (0)
#define PSTEP(k, j, t0) do {} while (0)
#define PARR(j, t0) do {} while (0)
#define PPROBE_WAVE 1
#endif

constexpr int PB = 16;                 // block edge
constexpr int PP = 18;                 // LDS pitch of a panel row (16 doubles + 2): rows 16-byte aligned, and the 16 lanes of a quarter wave
                                       // reading 32 bytes of 16 consecutive rows hit all 64 banks once (ds_read_b128, conflict-free)
constexpr int PROWS = 256;
constexpr int NW = 16;                 // waves of the panel workgroup: wave 0 carries the serial chain, waves 1..15 own the trailing tiles
constexpr int NSLOT = 10;              // tile slots per wave: 119 tiles (i, k), 1 <= k <= i <= 15 except (1,1), ~30 per SIMD, SIMD 0 with three tile waves
constexpr size_t kPanelLds = (size_t)(2 * PROWS * PP + 256 + 256 + 32 + 16 * PP) * sizeof(double);
// the four-wave form for fronts of at most 128 real columns in the panel (round 6): one chain wave + three tile waves, 128 panel rows
constexpr int PROWS4 = 128;
constexpr size_t kPanelLds4 = (size_t)(2 * PROWS4 * PP + 256 + PROWS4 + 32 + 16 * PP) * sizeof(double);

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL store to be
// acknowledged (s_waitcnt vmcnt(0): ~1 us per step here, where L / Dinv / y stream out while the factorisation goes on
// and nobody reads them back inside the kernel).
COV_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lane id from the hardware, inside a volatile asm: it is re-issued where it is used (two VALU instructions) instead of being
// computed once, hoisted out of the step loop and reloaded from scratch — a memory latency on the serial chain per reload
COV_DEV int hw_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// DPP row_newbcast (gfx90a+, the one DPP control 64-bit operations take): lane C of every 16-lane row to the whole row.
// Written as inline assembly because the compiler does not fold v_mov_b64_dpp into the multiply-add (a dependent pair per element, twice
// the instructions on the chain) — and therefore WITHOUT the compiler's hazard handling: a DPP source written by a VALU instruction needs
// two wait states (five after an EXEC write). diag_sweep keeps that distance by construction: the broadcast of the pivot carries its own
// s_nop, every multiply-add reads the register its predecessor of the PREVIOUS pivot wrote (>= 4 instructions earlier, asm volatile
// statements keep their order), and dpp_fence() separates the sweep from whatever wrote the registers before it.
template <int C> COV_DEV double dpp_bcast(double v) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(C));
  return d;
}
// acc += (acc of lane C of this row) * m
template <int C> COV_DEV void dpp_fmac(double& acc, double m) {
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(C));
}
COV_DEV void dpp_fence(double (&a)[16], double (&w)[4]) {
  asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
               "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}

// Pivots C .. 15 of the diagonal-block sweep of k_potrf_panel's wave 0 (see there), unrolled by recursion: the DPP control is an immediate
template <int C> COV_DEV void diag_sweep(double (&a)[16], double (&w)[4], int r, bool& bad, double& rsr) {
  if constexpr (C < 16) {
    int rr = r;
    asm volatile("" : "+v"(rr));  // (lane masks recomputed per pivot instead of sixteen hoisted SGPR pairs)
    const double d = dpp_bcast<C>(a[C]);
    bad = bad || !(d > 0.0);  // (a non-positive pivot poisons the block with inf/NaN: flagged, the caller discards the solve)
    // 1/d: hardware estimate + two Newton steps arranged as r1 = r + r e, r2 = r1 + r1 e^2 (e = 1 - d r): three dependent
    // operations after the estimate instead of four
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, r0, 1.0);
    const double r1 = fma(r0, e0, r0), e1 = e0 * e0;
    const double rinv = fma(r1, e1, r1);
    const double nt = -(a[C] * rinv);
#pragma unroll
    for (int cc = C + 1; cc < 16; ++cc) dpp_fmac<C>(a[cc], nt);
    const double ntw = (rr > C) ? nt : 0.0;  // rows <= C of W are final rows of the inverse
#pragma unroll
    for (int e = 0; e < 4; ++e) dpp_fmac<C>(w[e], ntw);
    rsr = (rr == C) ? d : rsr;   // the lane's own pivot
    diag_sweep<C + 1>(a, w, r, bad, rsr);
  }
}

// acc (tile (i,k), accumulator layout: row = (lane>>4) + 4 reg, col = lane & 15) -= P_i P_k^T for the 16-column panel `pan`
COV_DEV v4f64 tile_update(v4f64 acc, const double* pan, int i, int k, int fr, int fk) {
  const double* pa = pan + (PB * i + fr) * PP + 4 * fk;
  const double* pb = pan + (PB * k + fr) * PP + 4 * fk;
  const double2 a01 = *reinterpret_cast<const double2*>(pa), a23 = *reinterpret_cast<const double2*>(pa + 2);
  const double2 b01 = *reinterpret_cast<const double2*>(pb), b23 = *reinterpret_cast<const double2*>(pb + 2);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a01.x, b01.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a01.y, b01.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a23.x, b23.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a23.y, b23.y, acc, 0, 0, 0);
  return acc;
}

// the 119 trailing tiles (i, k), 1 <= k <= i <= 15 except (1,1), in column-major order, packed i | k << 8
struct PanelTileTab { int v[119]; };
constexpr PanelTileTab make_panel_tiles() {
  PanelTileTab t{};
  int u = 0;
  for (int k = 1; k < 16; ++k)
    for (int i = k; i < 16; ++i) { if (i == 1 && k == 1) continue; t.v[u++] = i | (k << 8); }
  return t;
}
__constant__ PanelTileTab kPanelTile = make_panel_tiles();
// the 27 trailing tiles of an 8-block panel (i, k), 1 <= k <= i <= 7 except (1,1), column-major (the four-wave form)
struct PanelTileTab8 { int v[27]; };
constexpr PanelTileTab8 make_panel_tiles8() {
  PanelTileTab8 t{};
  int u = 0;
  for (int k = 1; k < 8; ++k)
    for (int i = k; i < 8; ++i) { if (i == 1 && k == 1) continue; t.v[u++] = i | (k << 8); }
  return t;
}
__constant__ PanelTileTab8 kPanelTile8 = make_panel_tiles8();

// Panel j (LDS, complete: block column j of the factor from row o = 16 j on, raw diagonal block + its column factors `rs`) leaves for
// memory, and the right-hand side below it takes y_j (in sRhs[o ..)): thread u of NT, half rows of 64 bytes.
template <int NT>
COV_DEV void panel_out(const double* pan, double* sRhs, const double* rs, double* Mg, size_t ld, int o, int n, int u, double* yo) {
  for (int row = o + PB + (u >> 1); row < n; row += NT / 2) {
    const double2* src = reinterpret_cast<const double2*>(pan + row * PP + 8 * (u & 1));
    double2* dst = reinterpret_cast<double2*>(Mg + (size_t)row * ld + o + 8 * (u & 1));
    const double2 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
  }
  for (int e = NT - 1 - u; e < 256; e += NT) {  // the block's own rows: lower part, scaled by column (from the other end of the threads than the half rows)
    const int rr = e >> 4, cc = e & 15;
    if (cc <= rr) Mg[(size_t)(o + rr) * ld + o + cc] = pan[(o + rr) * PP + cc] * rs[cc];
  }
  for (int row = o + PB + (NT - 1 - u); row < n; row += NT) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
    for (int c = 0; c < PB; c += 4) {
      t0 += sRhs[o + c] * pan[row * PP + c]; t1 += sRhs[o + c + 1] * pan[row * PP + c + 1];
      t2 += sRhs[o + c + 2] * pan[row * PP + c + 2]; t3 += sRhs[o + c + 3] * pan[row * PP + c + 3];
    }
    sRhs[row] -= (t0 + t1) + (t2 + t3);
  }
  if (yo != nullptr && u < PB) yo[o + u] = sRhs[o + u];
}

// Factor the (16 nb)-order diagonal block at (k0, k0), nb = 16 (a 256-column panel) or fewer (the last panel of a front).
// Dinv_out: block j at Dinv_out + (j >> 3) * 128*128 + (j & 7) * 256, [16][16] row-major (zeros above the diagonal).
//
// Sixteen waves, two roles, each a loop of its own (a wave's registers are then either the sweep's or the tiles'), two
// workgroup barriers per block column j:
//   wave 0 (the chain)    (a) diagonal block j  ->  X(j+1,j) = T(j+1,j) Dinv_j^T  ->  T(j+1,j+1) -= X X^T  ->  (a) diagonal block j+1 ...
//   waves on SIMDs 1-3    (twelve; wave w runs on SIMD w & 3) own the 119 trailing 16x16 tiles in REGISTERS in the MFMA accumulator
//                         layout, dealt round-robin in column-major order: between barriers X(j) and Y(j) they solve the remaining tiles of
//                         panel j (i >= j+2) against Dinv_j; between Y(j) and X(j+1) — while wave 0 already factors block j+1 — they apply
//                         panel j to block column j+1 (handed on through LDS), to the diagonal tile (j+2,j+2) (handed to wave 0 through sDg
//                         one step ahead) and to the rest of their tiles. The matrix pipe runs one v_mfma_f64_16x16x4 in 64 cycles,
//                         dependent or not (tools/valu_probe.hip): 680 tile updates of 4 are 24 us of three matrix pipes, and the first
//                         eight steps are bound by them, the last eight by wave 0's chain.
//                         L and y leave ONE STEP BEHIND (panel_out), in the solve phase.
//   waves 4, 8, 12        (SIMD 0, beside the chain wave, which takes every issue slot there during its sweep) help to fill LDS and exit.
// The diagonal-block sweep of wave 0 is described at its place in the loop; the rest of the panel is X_ij = T_ij Dinv_j^T, 4 MFMAs per
// tile. (Round 1 found the product with a 128x128 explicit inverse too inaccurate for this system; a 16x16 block inverse formed by
// substitution is the standard blocked-TRSM building block and tests/test_gpu_parity.py::test_mfma_cholesky_ill_conditioned_blocks and
// the full-size parity tests hold with it.)
// Round 6: (i) every front factors only ITS OWN real 16-column blocks of the panel (`own`: the launch-wide nb is the widest front's; a front's
// further blocks are identity padding — L = I, block inverses = I, y = 0 are in place and stay), and a launch lists only the fronts that have
// a real column in this panel (`list`): on the 12-agent map a level mixes 1 000-2 000 fronts of 9 .. 700 unknowns, and every one of them ran all
// the steps of the widest in every panel of the level — 16 of that map's 38 ms per iteration. (ii) NWV = 4: the same kernel with ONE chain wave and
// THREE tile waves (27 trailing tiles of an 8-block panel, 128 panel rows, 42 KB of LDS) for fronts of at most 128 real columns in the panel — the
// speed-bias segments and the small pose fronts, thousands per level: three workgroups per CU instead of one. The arithmetic of a front is the
// same in both forms (same tile updates in the same order).
template <int NWV>
COV_DEV void potrf_panel_body(double* __restrict__ M, size_t ld, int k0, int nb, double* __restrict__ Dinv_out, int* flag,
                              const double* __restrict__ rhs, double* __restrict__ yout, size_t bsM, size_t bsL, size_t bsR,
                              const long long* __restrict__ btab, const int* __restrict__ own, const int* __restrict__ list, DevSignal sa, DevSignal sb) {
  constexpr int PROWS = (NWV == 16) ? 256 : PROWS4;   // (shadows the 16-wave constant)
  // records of the chain's stream that stand right in front of this launch are published by its first thread (CholAux::publish_handle): at this
  // point everything enqueued before the launch is complete — one launch less per panel on the serial chain
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (sa.flag != nullptr) __hip_atomic_store(sa.flag, sa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sb.flag != nullptr) __hip_atomic_store(sb.flag, sb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  constexpr int NTW = (NWV == 16) ? 12 : 3;           // tile waves
  const int front = list != nullptr ? list[blockIdx.x] : (int)blockIdx.x;
  if (own != nullptr) {
    const int real = __builtin_amdgcn_readfirstlane(own[front]) - k0;
    nb = min(nb, (real + PB - 1) / PB);
    if (nb <= 0) return;   // (workgroup-uniform) no real column of this front in the panel
  }
  if (btab != nullptr) { M += (size_t)btab[2 * front]; ld = (size_t)btab[2 * front + 1]; }  // fronts of unequal order (GemmArgs::btab)
  else M += (size_t)front * bsM;
  Dinv_out += (size_t)front * bsL;
  if (rhs != nullptr) { rhs += (size_t)front * bsR; yout += (size_t)front * bsR; }
  extern __shared__ __attribute__((aligned(16))) double sP[];  // panel[2][PROWS][PP] | sDv[16][16] | sRhs[PROWS] | sdd[2][16] | sDg[16][PP]
  double* sDv = sP + 2 * PROWS * PP;
  double* sRhs = sDv + 256;
  double* sdd = sRhs + PROWS;   // [2][16] 1/sqrt of the block's 16 pivots
  double* sDg = sdd + 32;       // [16][PP] the diagonal tile wave 0 takes over next (panels before the current one applied)
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int n = PB * nb;
  double* Mg = M + (size_t)k0 * ld + k0;
  const long long tp0 = PPROBE_T0();

  // ---- block column 0 (its diagonal tile SYMMETRIC: an entry above the diagonal is read from its mirror image — the sweep of wave 0
  // wants whole rows), the diagonal tile (1,1) and the right-hand side go to LDS: by the four waves of SIMD 0 (a row per thread), which
  // own no tiles. The tile waves only ISSUE the loads of their tiles before the barrier behind the prologue (first needed at the end of
  // step 0) and go on to barrier X(0): the chain wave starts its sweep one memory round trip after the launch, beside them.
  auto fill_lds = [&]() {
    const int fid = (NWV == 16) ? 64 * (wave >> 2) + (tid0 & 63) : tid0;   // 0 .. 255 (four-wave form: every wave fills, then the tile waves load their tiles)
    double fv[PB], frhs = 0.0, fdg = 0.0;
#pragma unroll
    for (int c = 0; c < PB; ++c) fv[c] = 0.0;
    if (fid < n) {
      if (fid >= PB) {
#pragma unroll
        for (int c = 0; c < PB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(Mg + (size_t)fid * ld + c); fv[c] = v.x; fv[c + 1] = v.y; }
      } else {
#pragma unroll
        for (int c = 0; c < PB; ++c) fv[c] = Mg[(size_t)(c > fid ? c : fid) * ld + (c > fid ? fid : c)];
      }
      if (rhs != nullptr) frhs = rhs[k0 + fid];
    }
    if (nb > 1) {  // diagonal tile (1,1): wave 0 takes it over at step 0 (see below)
      const int rr = fid >> 4, cc = fid & 15;
      fdg = Mg[(size_t)(PB + (cc <= rr ? rr : cc)) * ld + PB + (cc <= rr ? cc : rr)];
    }
    if (fid < n) {
#pragma unroll
      for (int c = 0; c < PB; c += 2) *reinterpret_cast<double2*>(sP + fid * PP + c) = double2{fv[c], fv[c + 1]};
    }
    if (fid < PROWS) sRhs[fid] = frhs;
    if (nb > 1) sDg[(fid >> 4) * PP + (fid & 15)] = fdg;
    // (no barrier of its own: the first one everybody meets is X(0); before it the chain wave only reads rows 0..15 of block column 0,
    //  which its own lanes have just written)
  };

  if (wave == 0) {
    // ================================================================ the chain
    PPROBE_DECL();
    fill_lds();
    PPROBE_ACC(4, tp0);
    __builtin_amdgcn_s_setprio(3);
    for (int j = 0; j < nb; ++j) {
      double* cur = sP + (j & 1) * PROWS * PP;          // block column j, rows 16 j .. n
      double* oth = sP + ((j + 1) & 1) * PROWS * PP;    // block column j+1 (being assembled)
      double* dv = sDv;
      const int o = PB * j;
      const long long tq0 = PPROBE_T0();
      // ---- (a): factor the diagonal block and form its inverse.
      // The sweep lives in REGISTERS, one matrix row per lane, and what a pivot needs of another row arrives by DPP row_newbcast (lane c
      // of every 16-lane row to the whole row) — no LDS round trip on the chain. Lane (r, g): a[0..15] = row r of the SYMMETRIC block (the
      // four 16-lane rows carry identical copies), w[0..3] = W[r][4g..4g+3] (W starts as I). Pivot c: d = a_c[c], t_r = a_r[c] / d (the
      // lane's own register), a_r[cc] -= t_r a_c[cc] for cc > c (both triangles: row c' must still be whole when its turn comes; rows <= c
      // turn into leftovers nobody reads), W_r: -= t_r W_c: for r > c — the forward substitution L X = I in outer-product form with the
      // same multipliers. L = A diag(d)^-1/2 and X = diag(d)^-1/2 W are scaled once at the end.
      const int ln = hw_lane_id();
      const int r = ln & 15, g = ln >> 4;
      double a[PB], w[4];
#pragma unroll
      for (int e = 0; e < PB; e += 2) { const double2 v = *reinterpret_cast<const double2*>(cur + (o + r) * PP + e); a[e] = v.x; a[e + 1] = v.y; }
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (4 * g + e == r) ? 1.0 : 0.0;
      bool bad = false;
      double rsr = 1.0;
      dpp_fence(a, w);
      diag_sweep<0>(a, w, r, bad, rsr);
      if (bad && ln == 0) atomicOr(flag, 1);
      // 1/sqrt(d_r), once per lane: X = diag(d)^-1/2 W (by row) here; L = A diag(d)^-1/2 (by column) is applied by the threads that store
      // L_jj (nobody reads it from LDS before): the raw columns and the sixteen factors go to LDS
      {
        const double dvv = (rsr > 0.0) ? rsr : 1.0;
        double rs = __builtin_amdgcn_rsq(dvv);
        rs = rs * (1.5 - 0.5 * dvv * rs * rs);
        rs = rs * (1.5 - 0.5 * dvv * rs * rs);
        rsr = rs;
      }
      if (ln < PB) {
        sdd[PB * (j & 1) + r] = rsr;   // (two buffers: the storing threads of step j run beside the sweep of step j + 1)
#pragma unroll
        for (int e = 0; e < PB; e += 2) *reinterpret_cast<double2*>(cur + (o + r) * PP + e) = double2{a[e], a[e + 1]};   // (above the diagonal: leftovers nobody reads)
      }
      double* dst = Dinv_out + (size_t)(j >> 3) * kTile * kTile + (size_t)(j & 7) * 256 + r * PB + 4 * g;
      double xv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[e] = (4 * g + e <= r) ? w[e] * rsr : 0.0;
        asm volatile("" : "+v"(xv[e]));  // selects, not a branch around the stores
        dv[r * PB + 4 * g + e] = xv[e];
        dst[e] = xv[e];
      }
      PPROBE_ACC(5, tq0);
      PSTEP(0, j, tp0);
      lds_barrier();  // ---- X(j): L_jj, Dinv_j in LDS; block column j complete in `cur`; trailing tiles carry panels < j
      PPROBE_ACC(1, tq0);
      PSTEP(1, j, tp0);
      const long long tq1 = PPROBE_T0();
      if (j + 1 < nb) {
        // X(j+1,j) = T(j+1,j) Dinv_j^T, formed TRANSPOSED — X^T = Dinv_j T^T, the operands of the same four instructions swapped (Dinv_j[r][4g..4g+3]
        // is what this lane holds) —: lane (r, g) then holds X[r][g], X[r][g+4], X[r][g+8], X[r][g+12], which IS an operand of the next diagonal
        // tile's update T(j+1,j+1) -= X X^T under the k-slot assignment (g, s) <-> column g + 4 s (any assignment does, taken for both
        // operands): no LDS round trip between the two products. The tile waves get X in row layout through `cur`, the next sweep the
        // updated tile through `oth`.
        double* tp = cur + (o + PB + r) * PP + 4 * g;
        const double2 ta = *reinterpret_cast<const double2*>(tp), tc = *reinterpret_cast<const double2*>(tp + 2);
        v4f64 dg;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) dg[rg] = sDg[(g + 4 * rg) * PP + r];
        v4f64 xt = v4f64{0.0, 0.0, 0.0, 0.0};
        xt = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[0], ta.x, xt, 0, 0, 0);
        xt = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[1], ta.y, xt, 0, 0, 0);
        xt = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[2], tc.x, xt, 0, 0, 0);
        xt = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[3], tc.y, xt, 0, 0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-xt[s2], xt[s2], dg, 0, 0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) cur[(o + PB + r) * PP + g + 4 * s2] = xt[s2];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) oth[(o + PB + g + 4 * rg) * PP + r] = dg[rg];
      }
      lds_barrier();  // ---- Y(j): panel j complete in `cur`, the next diagonal tile in `oth`
      PPROBE_ACC(2, tq1);
      PSTEP(2, j, tp0);
    }
    __builtin_amdgcn_s_setprio(0);
    PPROBE_FLUSH();
    return;
  }

  const int jsplit = (NWV == 16 && nb > 8) ? nb - 8 : 0;   // steps before it are bound by the matrix pipes (the chain wave waits), the last eight by the chain
  if (NWV == 16 && (wave & 3) == 0) {
    // ================================================================ SIMD 0's other waves: y_j, and the way out of panel j while the steps
    // are bound by the matrix pipes (the chain wave, which takes every issue slot of this SIMD during its sweep, then waits for the tiles)
    fill_lds();
    const int lq = tid0 & 63;
    for (int j = 0; j < nb; ++j) {
      const double* cur = sP + (j & 1) * PROWS * PP;
      const int o = PB * j;
      PARR(j, tp0);
      lds_barrier();  // ---- X(j)
      if (wave == 4 && lq < PB) {  // y_j = Dinv_j b_j
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
#pragma unroll
        for (int k = 0; k < PB; k += 4) {
          y0 += sDv[lq * PB + k] * sRhs[o + k]; y1 += sDv[lq * PB + k + 1] * sRhs[o + k + 1];
          y2 += sDv[lq * PB + k + 2] * sRhs[o + k + 2]; y3 += sDv[lq * PB + k + 3] * sRhs[o + k + 3];
        }
        sRhs[o + lq] = (y0 + y1) + (y2 + y3);
      }
      lds_barrier();  // ---- Y(j)
      if (j < jsplit) panel_out<192>(cur, sRhs, sdd + PB * (j & 1), Mg, ld, o, n, 64 * ((wave >> 2) - 1) + lq, yout != nullptr ? yout + k0 : nullptr);
      if (wave == PPROBE_WAVE) PSTEP(3, j, tp0);
    }
    return;
  }


  // ================================================================ the tile waves
  PPROBE_DECL();
  const int lane0 = tid0 & 63, fr0 = lane0 & 15, fk0 = lane0 >> 4;
  const int tw = (NWV == 16) ? 3 * (wave >> 2) + (wave & 3) - 1 : wave - 1;   // 0 .. NTW-1
  if (NWV != 16) fill_lds();
  // ---- tile slots of this wave (wave-uniform, packed i | k << 8; k = 99: none): the 119 tiles (i, k), 1 <= k <= i <= 15 except (1,1), in
  // column-major order, round-robin: every SIMD carries the same number of tiles at every step
  int tik[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int u = NTW * s + tw;
    constexpr int NTILE = (NWV == 16) ? 119 : 27;
    const int pk = (NWV == 16) ? kPanelTile.v[u < NTILE ? u : 0] : kPanelTile8.v[u < NTILE ? u : 0];   // (a table: the search for the column was ~1 us of scalar loops per wave)
    tik[s] = (u < NTILE && (pk & 255) < nb) ? pk : (99 << 8);
  }
  // ---- the wave's trailing tiles into registers: the loads are only ISSUED here (see above): addresses from four per-lane offsets (and
  // four mirrored ones for the diagonal tiles, kept symmetric) and a wave-uniform base per tile; a slot without a tile loads tile (1,1)
  // and never looks at it again (a select would wait for the load)
  unsigned offn[4], offm[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int rw = fk0 + 4 * rg;
    offn[rg] = (unsigned)rw * (unsigned)ld + (unsigned)fr0;
    offm[rg] = fr0 > rw ? (unsigned)fr0 * (unsigned)ld + (unsigned)rw : offn[rg];
  }
  v4f64 acc[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const bool on = (tik[s] >> 8) != 99;
    const int ii = on ? (tik[s] & 255) : 1, kk = on ? (tik[s] >> 8) : 1;
    const double* base = Mg + (size_t)(PB * ii) * ld + PB * kk;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) acc[s][rg] = base[ii == kk ? offm[rg] : offn[rg]];
  }
  // (The first step is as long as the tile loads take to issue, ~5 us: 244 KB into ONE CU at ~50 GB/s. Measured and dropped: half of them
  //  behind barrier X(0) — the conditional load inside the step loop costs 38 spilled registers —; 16-byte loads by lane pairs with a DPP
  //  swap behind X(0): half the instructions, the same time.)
  for (int j = 0; j < nb; ++j) {
    double* cur = sP + (j & 1) * PROWS * PP;
    double* oth = sP + ((j + 1) & 1) * PROWS * PP;
    const double* dv = sDv;
    // every per-lane constant of the step is rebuilt from the hardware lane id here: kept across the loop they are spilled
    // (the accumulator tiles take the registers) and every reload is a memory latency
    const int lq = hw_lane_id();
    const int fr = lq & 15, fk = lq >> 4;
    PARR(j, tp0);
    lds_barrier();  // ---- X(j)
    if (NWV != 16 && tw == 0 && lq < PB) {  // y_j = Dinv_j b_j (the 16-wave form: wave 4)
      const int o = PB * j;
      double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
#pragma unroll
      for (int k = 0; k < PB; k += 4) {
        y0 += sDv[lq * PB + k] * sRhs[o + k]; y1 += sDv[lq * PB + k + 1] * sRhs[o + k + 1];
        y2 += sDv[lq * PB + k + 2] * sRhs[o + k + 2]; y3 += sDv[lq * PB + k + 3] * sRhs[o + k + 3];
      }
      sRhs[o + lq] = (y0 + y1) + (y2 + y3);
    }
    for (int i = j + 2 + tw; i < nb; i += NTW) {  // X(i,j) for i >= j+2
      double* tp = cur + (PB * i + fr) * PP + 4 * fk;
      const double* dp = dv + fr * PB + 4 * fk;
      const double2 ta = *reinterpret_cast<const double2*>(tp), tc = *reinterpret_cast<const double2*>(tp + 2);
      const double2 tb = *reinterpret_cast<const double2*>(dp), td = *reinterpret_cast<const double2*>(dp + 2);
      v4f64 x = v4f64{0.0, 0.0, 0.0, 0.0};
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(ta.x, tb.x, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(ta.y, tb.y, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(tc.x, td.x, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(tc.y, td.y, x, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) cur[(PB * i + fk + 4 * rg) * PP + fr] = x[rg];
    }
    lds_barrier();  // ---- Y(j): panel j complete in `cur`, the next diagonal tile in `oth`
    const long long tq2 = PPROBE_T0();
    // panel j onto every trailing tile of this wave; block column j+1 goes on to `oth` and the diagonal tile (j+2,j+2) to sDg on the way.
    // Tile (j+1,j+1) is wave 0's already.
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      int pk = tik[s];
      asm volatile("" : "+s"(pk));  // keeps the LDS row addresses of all slots from being hoisted out of the j loop (spills)
      const int i = pk & 255, k = pk >> 8;
      if (k >= j + 1 && k != 99 && !(k == j + 1 && i == k)) {
        acc[s] = tile_update(acc[s], cur, i, k, fr, fk);
        if (k == j + 1 || (k == j + 2 && i == k)) {
          double* dstp = (i == k) ? sDg + fk * PP + fr : oth + (PB * i + fk) * PP + fr;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) dstp[4 * rg * PP] = acc[s][rg];
        }
      }
    }
    PPROBE_ACC(6, tq2);
    if (wave == PPROBE_WAVE) PSTEP(3, j, tp0);
    const long long tq3 = PPROBE_T0();
    if (j >= jsplit) panel_out<64 * NTW>(cur, sRhs, sdd + PB * (j & 1), Mg, ld, PB * j, n, 64 * tw + lq, yout != nullptr ? yout + k0 : nullptr);   // (the tile waves have the time now)
    PPROBE_ACC(7, tq3);
  }
  PPROBE_FLUSH();
}

__global__ __launch_bounds__(64 * NW) void k_potrf_panel(double* __restrict__ M, size_t ld, int k0, int nb, double* __restrict__ Dinv_out, int* flag,
                                                         const double* __restrict__ rhs, double* __restrict__ yout, size_t bsM, size_t bsL, size_t bsR,
                                                         const long long* __restrict__ btab, const int* __restrict__ own, const int* __restrict__ list, DevSignal sa, DevSignal sb) {
  potrf_panel_body<16>(M, ld, k0, nb, Dinv_out, flag, rhs, yout, bsM, bsL, bsR, btab, own, list, sa, sb);
}
// (three workgroups per CU: 42 KB of LDS each; at most 168 registers per wave)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_potrf_panel4(double* __restrict__ M, size_t ld, int k0, int nb, double* __restrict__ Dinv_out, int* flag,
                                                         const double* __restrict__ rhs, double* __restrict__ yout, size_t bsM, size_t bsL, size_t bsR,
                                                         const long long* __restrict__ btab, const int* __restrict__ own, const int* __restrict__ list, DevSignal sa, DevSignal sb) {
  potrf_panel_body<4>(M, ld, k0, nb, Dinv_out, flag, rhs, yout, bsM, bsL, bsR, btab, own, list, sa, sb);
}

struct TrsmSubArgs {
  double* M; size_t ld;
  int k0;                 // first column of the panel
  int r0;                 // first row (multiple of 16); workgroup x handles rows r0 + 16 x ..
  const double* Dinv;     // block inverses of the panel's first tile (second tile 128*128 further)
  double* rhs; const double* yvec;   // forward substitution riding along: rhs[rows] -= X[rows, :] y[k0 ..)
  size_t bsM, bsL, bsR;
  const int* live; int tI;           // see GemmArgs (k_chol.hip)
  int chain;                         // 1: launched on the serial chain — its waves raise their issue priority over the bulk update's
  const long long* btab;             // see GemmArgs (k_chol.hip)
  const int* own;                    // [batch] real interior order of every front (nullptr: the launch-wide NB applies to all)
};

// X = A L^-T on a 16-row slab, L = the (16 NB)-order factor at (k0, k0). One wave, everything in registers.
// (Measured and dropped: FOUR waves per slab for the 16 slabs of the next panel's rows on the serial chain — block columns dealt
//  to the waves, Z_j handed on through LDS with one LDS-only barrier per step. Correct, but 260 VGPRs once unrolled — the
//  workgroup then only starts on an empty CU — and 3.01 vs 2.95 ms for the whole solve.)
// acc[i][reg] at lane (n = lane & 15, fk = lane >> 4) holds Z[16 i + 4 fk + reg][n] = X[row0 + n][k0 + 16 i + 4 fk + reg].
template <int NB>
COV_DEV void trsm_sub_body(const TrsmSubArgs& g) {
  const int batch = blockIdx.y, lane = threadIdx.x, n = lane & 15, fk = lane >> 4;
  const int row0 = g.r0 + PB * (int)blockIdx.x;
  if (g.chain) __builtin_amdgcn_s_setprio(3);  // resident beside bulk-update waves that keep the matrix pipe busy: without it the slab runs 2.5x longer
  if (g.live != nullptr) {
    const int nI = g.live[2 * batch], nO = g.live[2 * batch + 1];
    const int tp = g.k0 / kTile, tr = row0 / kTile;
    if (!(tp < nI || (tp >= g.tI && tp - g.tI < nO))) return;
    if (!(tr < nI || (tr >= g.tI && tr - g.tI < nO))) return;
  }
  double* Mb = g.M + (g.btab != nullptr ? (size_t)g.btab[2 * batch] : (size_t)batch * g.bsM);
  const size_t ld = g.btab != nullptr ? (size_t)g.btab[2 * batch + 1] : g.ld;
  const double* Db = g.Dinv + (size_t)batch * g.bsL;
  const int pr = 4 * (n & 3) + (n >> 2);  // logical row carried by A-operand lane n
  double* Arow = Mb + (size_t)(row0 + n) * ld + g.k0 + 4 * fk;
  const double* Lrow = Mb + (size_t)(g.k0 + pr) * ld + g.k0 + 4 * fk;
  const double* Drow = Db + pr * PB + 4 * fk;
  auto Ltile = [&](int i, int j) { return *reinterpret_cast<const v4f64*>(Lrow + (size_t)(PB * i) * ld + PB * j); };
  auto Dblk = [&](int j) { return *reinterpret_cast<const v4f64*>(Drow + (size_t)(j >> 3) * kTile * kTile + (j & 7) * 256); };
  v4f64 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = *reinterpret_cast<const v4f64*>(Arow + PB * i);
  // The L tiles stream through a small ring of registers, fetched RING-1 tiles ahead of their use across block-column
  // boundaries (every index below is a compile-time constant once the loops are unrolled). With a whole block column
  // double-buffered the kernel needed 416 VGPRs: a wave then only starts on a SIMD that is EMPTY, and on the serial
  // chain it queued behind the bulk update's workgroups (18 us alone, 52 us average in the run). ~200 fit beside one.
  constexpr int RING = 5;
  v4f64 ring[RING], dcur = Dblk(0), dnxt = dcur;
  int pi = 1, pj = 0, pt = 0;  // prefetch cursor: next tile (pi, pj) to fetch goes to ring[pt % RING]
#pragma unroll
  for (int k = 0; k < RING - 1; ++k)
    if (pj < NB - 1) { ring[pt % RING] = Ltile(pi, pj); ++pt; if (++pi >= NB) { ++pj; pi = pj + 1; } }
  int t = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j + 1 < NB) dnxt = Dblk(j + 1);
    v4f64 Z = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(dcur[s], acc[j][s], Z, 0, 0, 0);
    acc[j] = Z;
    const v4f64 Zn = -Z;
#pragma unroll
    for (int i = j + 1; i < NB; ++i) {
      if (pj < NB - 1) { ring[pt % RING] = Ltile(pi, pj); ++pt; if (++pi >= NB) { ++pj; pi = pj + 1; } }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[t % RING][s], Zn[s], acc[i], 0, 0, 0);
      ++t;
    }
    dcur = dnxt;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<v4f64*>(Arow + PB * i) = acc[i];
  if (g.rhs != nullptr) {  // rhs[row0 + n] -= sum_k X[n][k] y[k]: lane partial, fixed butterfly over the four lanes sharing n
    const double* yv = g.yvec + (size_t)batch * g.bsR + g.k0 + 4 * fk;
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const v4f64 y4 = *reinterpret_cast<const v4f64*>(yv + PB * i);
#pragma unroll
      for (int s = 0; s < 4; ++s) part += acc[i][s] * y4[s];
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (fk == 0) g.rhs[(size_t)batch * g.bsR + row0 + n] -= part;
  }
}

// The launch is sized for the widest front of the batch; a front with fewer real columns in this panel runs the shorter body
// (its further columns are identity padding: X = A = 0 there), one without any returns at once.
template <int NB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_trsm_sub(TrsmSubArgs g) {
  if (g.own != nullptr) {
    const int real = g.own[blockIdx.y] - g.k0;
    if (real <= 0) return;
    const int nbf = (real + PB - 1) / PB;
    if (NB > 4 && nbf <= 4) { trsm_sub_body<4>(g); return; }
    if (NB > 8 && nbf <= 8) { trsm_sub_body<8>(g); return; }
    if (NB > 12 && nbf <= 12) { trsm_sub_body<12>(g); return; }
  }
  trsm_sub_body<NB>(g);
}

// The same substitution with FOUR waves per slab, one per SIMD: the single wave above is bound by ITS matrix pipe (544 v_mfma_f64_16x16x4 of
// 64 cycles, dependent or not: 14.5 us before any load latency), and on the serial chain sixteen slabs are all there is to run. Wave w owns
// the blocks i = w, w + 4, ... of Z. Nobody waits at a barrier: block j of Z goes through LDS (all sixteen kept) behind a counter the
// other waves poll, and the wave that owns block j + 1 applies L(j+1, j) to it FIRST, publishes Z_{j+1} = Dinv_{j+1} acc_{j+1} and only
// then catches up on its other blocks — the chain is (read Z_j, 4 MFMAs, 4 MFMAs, write Z_{j+1}) per step, everything else fills in behind.
// The L tiles of a wave stream through a ring of registers in a FIXED order (step-major; its slot of the next owner's block row is fetched
// and dropped when it lies on or above the diagonal: ring positions stay compile-time constants).
#ifdef COVGPU_PROBE
__device__ long long g_tprobe[4][24];
#define TPROBE(k) do { if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_tprobe[wave][k] = clock64() - tstart; } while (0)
#else
#define TPROBE(k) do {} while (0)
#endif
template <int NB>
COV_DEV void trsm_sub4_body(const TrsmSubArgs& g, v4f64 (*zall)[64], int* zcount, double (*spart)[16]) {
  constexpr int MB = NB / 4;
  const int batch = blockIdx.y, lane = threadIdx.x & 63, n = lane & 15, fk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = g.r0 + PB * (int)blockIdx.x;
#ifdef COVGPU_PROBE
  const long long tstart = clock64();
#endif
  if (g.chain) __builtin_amdgcn_s_setprio(3);
  double* Mb = g.M + (g.btab != nullptr ? (size_t)g.btab[2 * batch] : (size_t)batch * g.bsM);
  const size_t ld = g.btab != nullptr ? (size_t)g.btab[2 * batch + 1] : g.ld;
  const double* Db = g.Dinv + (size_t)batch * g.bsL;
  const int pr = 4 * (n & 3) + (n >> 2);  // logical row carried by A-operand lane n
  double* Arow = Mb + (size_t)(row0 + n) * ld + g.k0 + 4 * fk;
  const double* Lrow = Mb + (size_t)(g.k0 + pr) * ld + g.k0 + 4 * fk;
  const double* Drow = Db + pr * PB + 4 * fk;
  auto Ltile = [&](int i, int j) { return *reinterpret_cast<const v4f64*>(Lrow + (size_t)(PB * i) * ld + PB * j); };
  auto Dblk = [&](int j) { return *reinterpret_cast<const v4f64*>(Drow + (size_t)(j >> 3) * kTile * kTile + (j & 7) * 256); };
  v4f64 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = *reinterpret_cast<const v4f64*>(Arow + PB * (4 * m + wave));
  v4f64 dmine[MB];   // the block inverses of this wave's blocks
#pragma unroll
  for (int m = 0; m < MB; ++m) dmine[m] = Dblk(4 * m + wave);
  constexpr int RING = 8;
  v4f64 ring[RING];
  // fetch cursor over the fixed sequence: for j = 0 .. NB-2, for m = (j + 1) / 4 .. MB-1 -> tile (4 m + wave, j)
  int pj = 0, pm = 0, pt = 0;
  auto fetch = [&]() {
    if (pj < NB - 1) {
      ring[pt % RING] = Ltile(4 * pm + wave, pj); ++pt;
      if (++pm >= MB) { ++pj; pm = (pj + 1) >> 2; }
    }
  };
#pragma unroll
  for (int k = 0; k < RING - 1; ++k) fetch();
  auto publish = [&](int j, int m) {   // Z_j = Dinv_j acc_j: final block of X, and -Z_j for everybody
    v4f64 Z = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(dmine[m][s], acc[m][s], Z, 0, 0, 0);
    acc[m] = Z;
    zall[j][lane] = -Z;
    if (lane == 0) __hip_atomic_store(zcount, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of a wave execute in order)
  };
  if (wave == 0) publish(0, 0);
  int t = 0;
#pragma unroll
  for (int j = 0; j + 1 < NB; ++j) {
    const int o1 = (j + 1) & 3, m1 = (j + 1) >> 2;
    if (wave != (j & 3)) {   // (the publisher itself need not look)
      int spins = 0;
      while (__hip_atomic_load(zcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < j + 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    const v4f64 Zn = zall[j][lane];
    if (NB == 16) TPROBE(j);
#pragma unroll
    for (int m = m1; m < MB; ++m) {
      fetch();
      if (m > m1 || wave >= o1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[t % RING][s], Zn[s], acc[m], 0, 0, 0);
      }
      ++t;
      if (m == m1 && wave == o1) publish(j + 1, m1);
    }
  }
  if (NB == 16) TPROBE(16);
#pragma unroll
  for (int m = 0; m < MB; ++m) *reinterpret_cast<v4f64*>(Arow + PB * (4 * m + wave)) = acc[m];
  if (g.rhs != nullptr) {  // rhs[row0 + n] -= sum_k X[n][k] y[k]: lane partial, fixed butterfly over the four lanes sharing n, waves in order
    const double* yv = g.yvec + (size_t)batch * g.bsR + g.k0 + 4 * fk;
    double part = 0.0;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const v4f64 y4 = *reinterpret_cast<const v4f64*>(yv + PB * (4 * m + wave));
#pragma unroll
      for (int s = 0; s < 4; ++s) part += acc[m][s] * y4[s];
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (fk == 0) spart[wave][n] = part;
    lds_barrier();
    if (wave == 0 && fk == 0) g.rhs[(size_t)batch * g.bsR + row0 + n] -= ((spart[0][n] + spart[1][n]) + spart[2][n]) + spart[3][n];
  }
}

template <int NB>
__global__ __launch_bounds__(256) void k_trsm_sub4(TrsmSubArgs g) {
  __shared__ v4f64 zall[16][64];
  __shared__ double spart[4][16];
  __shared__ int zcount;
  if (g.live != nullptr) {
    const int batch = blockIdx.y, row0 = g.r0 + PB * (int)blockIdx.x;
    const int nI = g.live[2 * batch], nO = g.live[2 * batch + 1];
    const int tp = g.k0 / kTile, tr = row0 / kTile;
    if (!(tp < nI || (tp >= g.tI && tp - g.tI < nO))) return;
    if (!(tr < nI || (tr >= g.tI && tr - g.tI < nO))) return;
  }
  int nbf = NB;
  if (g.own != nullptr) {
    const int real = g.own[blockIdx.y] - g.k0;
    if (real <= 0) return;
    nbf = (real + PB - 1) / PB;
  }
  if (threadIdx.x == 0) zcount = 0;
  __syncthreads();
  if (NB > 4 && nbf <= 4) { trsm_sub4_body<4>(g, zall, &zcount, spart); return; }
  if (NB > 8 && nbf <= 8) { trsm_sub4_body<8>(g, zall, &zcount, spart); return; }
  if (NB > 12 && nbf <= 12) { trsm_sub4_body<12>(g, zall, &zcount, spart); return; }
  trsm_sub4_body<NB>(g, zall, &zcount, spart);
}

// (Measured and dropped, round 4 — both bit-identical to this kernel: (i) four slabs per workgroup with the factor streamed through LDS
//  by block columns, two block columns ahead: 36 us instead of 25 on the chain, a two-step prefetch distance does not cover a memory
//  latency and every step then pays one; (ii) this kernel with a ring of TWENTY tiles and one wave per SIMD: no change at all — the slab
//  is bound by its 544 dependent-issue MFMAs on one SIMD (16 Z chains of four dependent instructions each), not by the tile loads.)
// (Measured and dropped, round 4: a panel PIPELINE for multi-panel fronts — the factorisation publishing its block columns through
//  device-scope flags, a second launch on another stream following it in lock step with the substitution of the next panel's rows and
//  the update of the next diagonal block. Correct, but a root panel took ~250 us instead of 159: a device-scope load / store
//  acknowledgement costs ~4 us here, as long as a whole factorisation step. profiles/r04_ab_experiments.txt; the code is in the history
//  of this file, commit 5ecf8c0.)

// Backward substitution step for tile p with the 16x16 block inverses (Dinv == nullptr: x_p is given):
//   x_p = L_pp^-T y_p ; y[cols left of the tile] -= L[tile rows, cols]^T x_p.
// Thread c < 128 owns unknown c: every entry of L_pp it will need (rows of the blocks below its own) is loaded up front
// (one memory latency), then eight block steps of two barriers each. Every workgroup of the launch repeats this (cheaper
// than a separate launch on a launch-bound chain); the column update is split over the workgroups as before.
__global__ __launch_bounds__(256) void k_bwd_step_sub(const double* __restrict__ M, size_t ld, int p, const double* __restrict__ Dinv,
                                                       double* __restrict__ y, double* __restrict__ x, int ncol, size_t bsM, size_t bsL, size_t bsR,
                                                       const long long* __restrict__ btab, const int* __restrict__ live, int tI, BwdXfer xf) {
  if (live != nullptr) {  // padding tile of this front: x_p = 0 contributes nothing (GemmArgs::live)
    const int nI = live[2 * blockIdx.y], nO = live[2 * blockIdx.y + 1];
    if (!(p < nI || (p >= tI && p - tI < nO))) return;
  }
  if (btab != nullptr) { M += (size_t)btab[2 * blockIdx.y]; ld = (size_t)btab[2 * blockIdx.y + 1]; }
  else M += (size_t)blockIdx.y * bsM;
  y += (size_t)blockIdx.y * bsR; x += (size_t)blockIdx.y * bsR;
  if (Dinv != nullptr) Dinv += (size_t)blockIdx.y * bsL;
  __shared__ double sx[kTile];
  __shared__ double sv[kTile];
  __shared__ double part[8][33];
  const int tid = threadIdx.x, k0 = p * kTile;
  if (Dinv == nullptr) {
    if (tid < kTile) sx[tid] = x[k0 + tid];
    __syncthreads();
  } else {
    const int c = tid & 127, jbc = c >> 4, cl = c & 15;
    const bool act = tid < kTile;
    double v = act ? y[k0 + c] : 0.0;
    double dv[PB], Lc[7][PB];
#pragma unroll
    for (int r = 0; r < PB; ++r) dv[r] = act ? Dinv[jbc * 256 + r * PB + cl] : 0.0;
#pragma unroll
    for (int jb = 1; jb < 8; ++jb)
#pragma unroll
      for (int r = 0; r < PB; ++r) Lc[jb - 1][r] = (act && jbc < jb) ? M[(size_t)(k0 + PB * jb + r) * ld + k0 + c] : 0.0;
#pragma unroll
    for (int jb = 7; jb >= 0; --jb) {
      if (act && jbc == jb) sv[c] = v;
      __syncthreads();
      if (act && jbc == jb) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int r = 0; r < PB; r += 4) {
          s0 += dv[r] * sv[PB * jb + r]; s1 += dv[r + 1] * sv[PB * jb + r + 1];
          s2 += dv[r + 2] * sv[PB * jb + r + 2]; s3 += dv[r + 3] * sv[PB * jb + r + 3];
        }
        sx[c] = (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
      if (jb > 0 && act && jbc < jb) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int r = 0; r < PB; r += 4) {
          s0 += Lc[jb > 0 ? jb - 1 : 0][r] * sx[PB * jb + r]; s1 += Lc[jb > 0 ? jb - 1 : 0][r + 1] * sx[PB * jb + r + 1];
          s2 += Lc[jb > 0 ? jb - 1 : 0][r + 2] * sx[PB * jb + r + 2]; s3 += Lc[jb > 0 ? jb - 1 : 0][r + 3] * sx[PB * jb + r + 3];
        }
        v -= (s0 + s1) + (s2 + s3);
      }
    }
    if (blockIdx.x == 0 && tid < kTile) x[k0 + tid] = sx[tid];
    if (xf.gidx != nullptr && blockIdx.x == 0) {  // last step of a multifrontal front: its own unknowns -> solution vector
      const int node = xf.first + blockIdx.y, n = xf.own_dims[node];
      const int* gi = xf.gidx + xf.own_g[node];
      for (int i = tid; i < n; i += 256) xf.x[gi[i]] = (i < kTile) ? sx[i] : x[i];   // (tile 0 from LDS: p == 0 here)
    }
  }
  const int cl = tid & 31, rg = tid >> 5;
  const int col = blockIdx.x * 32 + cl;
  double acc = 0.0;
  if (col < ncol) {
    const double* Lc2 = M + (size_t)(k0 + 16 * rg) * ld + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += Lc2[(size_t)r * ld] * sx[16 * rg + r];
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

// Backward substitution, all GIVEN tile rows of a front at once (the ancestors' unknowns of a multifrontal front, the border
// of an arrow block): y[c] -= sum_{r in [r0, r1)} L[r][c] x[r] for the factored columns c < ncol. One workgroup per 128
// columns; its four waves take every fourth row (a row segment = 1 KiB, 16 rows in flight per wave), partial sums are added in
// wave order: deterministic. Replaces one launch per given tile (12 dependent launches for a 1.5k-row border).
__global__ __launch_bounds__(256) void k_bwd_given(const double* __restrict__ M, size_t ld, int r0, int r1, double* __restrict__ y,
                                                    const double* __restrict__ x, int ncol, size_t bsM, size_t bsR, const long long* __restrict__ btab,
                                                    const int* __restrict__ live, int tI, BwdXfer xf) {
  const int batch = blockIdx.y;
  if (live != nullptr) {  // rows beyond the front's real border are padding (x = 0); a front without interior columns has nothing to update
    r1 = min(r1, (tI + live[2 * batch + 1]) * kTile);
    if (live[2 * batch] == 0) return;
  }
  if (btab != nullptr) { M += (size_t)btab[2 * batch]; ld = (size_t)btab[2 * batch + 1]; }
  else M += (size_t)batch * bsM;
  y += (size_t)batch * bsR; x += (size_t)batch * bsR;
  __shared__ double2 part[4][64];
  extern __shared__ double sxg[];  // [r1 - r0] the given unknowns, staged once (a gather through the index list inside the row loop doubled its latency)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 128 + 2 * lane;
  // multifrontal front: the given unknowns come straight from the solution vector (row r0 + i = border scalar i of the front)
  if (xf.gidx != nullptr) {
    const int node = xf.first + batch;
    const int* gi = xf.gidx + xf.st_g[node];
    r1 = min(r1, r0 + xf.st_dims[node]);
    for (int i = threadIdx.x; i < r1 - r0; i += 256) sxg[i] = xf.x[gi[i]];
  } else {
    for (int i = threadIdx.x; i < r1 - r0; i += 256) sxg[i] = x[r0 + i];
  }
  __syncthreads();
  double2 acc = {0.0, 0.0};
  if (col < ncol) {
    const double* Lp = M + col;
    int r = r0 + wv;
    for (; r + 28 < r1; r += 32) {
      double2 v[8]; double xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = *reinterpret_cast<const double2*>(Lp + (size_t)(r + 4 * u) * ld); xv[u] = sxg[r + 4 * u - r0]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x * xv[u]; acc.y += v[u].y * xv[u]; }
    }
    for (; r < r1; r += 4) { const double2 v = *reinterpret_cast<const double2*>(Lp + (size_t)r * ld); const double xv = sxg[r - r0]; acc.x += v.x * xv; acc.y += v.y * xv; }
  }
  part[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && col < ncol) {
    const double2 a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
    y[col] -= ((a.x + b.x) + c.x) + d.x;
    y[col + 1] -= ((a.y + b.y) + c.y) + d.y;
  }
}

// Backward substitution of a whole multifrontal front in ONE launch (fronts of at most four interior tiles: every level below
// the top of the tree). Workgroup (column tile ct, row chunk k) of a front forms the partial sums of
//   y[c] -= sum_r L[r][c] x[r]   over ITS 256 given rows (the ancestors' unknowns, read straight from the solution vector)
// for the 128 interior columns of tile ct and parks them in `scr`; the workgroup of the front that finishes LAST (ticket
// counter; nobody waits for anybody: no co-residency assumption) adds them in chunk order and solves the interior tiles from
// the last to the first: x_p = L_pp^-T y_p with the block inverses, y[cols left of the tile] -= L[tile rows, cols]^T x_p.
// Replaces 1 + (interior tiles) dependent launches per level whose first streamed a whole border through one workgroup per
// column tile (latency-bound: ~50 us for a 1.5k-row border).
__global__ __launch_bounds__(256) void k_bwd_front(const double* __restrict__ M, int tI, int nchunk, double* __restrict__ y, const double* __restrict__ Dinv_all,
                                                    size_t bsL, size_t bsR, const long long* __restrict__ btab, const int* __restrict__ live, BwdXfer xf,
                                                    int* __restrict__ cnt, double* __restrict__ scr) {
  const int batch = blockIdx.y, tid = threadIdx.x;
  const int ct = blockIdx.x / nchunk, k = blockIdx.x % nchunk;
  const int nIt = live[2 * batch];            // real interior tiles of this front
  const int node = xf.first + batch;
  const int nst = xf.st_dims[node];
  const int nch = max(1, (nst + 255) / 256);  // row chunks of this front
  if (ct >= nIt || k >= nch) return;
  M += (size_t)btab[2 * batch];
  const size_t ld = (size_t)btab[2 * batch + 1];
  y += (size_t)batch * bsR;
  const double* Dinv_f = Dinv_all + (size_t)batch * bsL;
  double* scr_f = scr + (size_t)batch * gridDim.x * kTile;   // [tile][chunk][128]
  __shared__ double2 part[4][64];
  __shared__ int s_ticket;
  __shared__ double sx[kTile], sv[kTile], sxg[256];
  {
    const int r0 = tI * kTile + 256 * k, nr = min(256, nst - 256 * k);   // this chunk's rows [r0, r0 + nr)
    const int* gi = xf.gidx + xf.st_g[node] + 256 * k;
    if (tid < nr) sxg[tid] = xf.x[gi[tid]];
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int col = ct * kTile + 2 * lane;
    double2 acc = {0.0, 0.0};
    const double* Lp = M + (size_t)r0 * ld + col;
    int r = wv;
    for (; r + 28 < nr; r += 32) {
      double2 v[8]; double xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = *reinterpret_cast<const double2*>(Lp + (size_t)(r + 4 * u) * ld); xv[u] = sxg[r + 4 * u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x * xv[u]; acc.y += v[u].y * xv[u]; }
    }
    for (; r < nr; r += 4) { const double2 v = *reinterpret_cast<const double2*>(Lp + (size_t)r * ld); const double xv = sxg[r]; acc.x += v.x * xv; acc.y += v.y * xv; }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) {
      const double2 a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
      double2 t; t.x = ((a.x + b.x) + c.x) + d.x; t.y = ((a.y + b.y) + c.y) + d.y;
      double* dst = scr_f + ((size_t)ct * nchunk + k) * kTile + 2 * lane;
      __hip_atomic_store(dst, t.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, t.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- ticket: the last workgroup of the front goes on. The partial sums travel through device-scope accesses (they bypass the
  //      per-XCD L2, which is not coherent across XCDs) and are complete (vmcnt) before the ticket is drawn; a device-scope FENCE
  //      here would write back and INVALIDATE the XCD's whole L2 under every workgroup that still streams its rows of L (measured:
  //      the sweep got slower than the separate launches).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a workgroup-scope release fence does not wait for the write-through stores)
  __syncthreads();
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(&cnt[batch], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != nIt * nch - 1) return;
  if (tid == 0) __hip_atomic_store(&cnt[batch], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (next launch: ordered by the stream)
  for (int p = nIt - 1; p >= 0; --p) {
    const int k0 = p * kTile;
    const double* Dinv = Dinv_f + (size_t)p * kTile * kTile;
    {
      const int c = tid & 127, jbc = c >> 4, cl = c & 15;
      const bool act = tid < kTile;
      double v = act ? y[k0 + c] : 0.0;
      // (eight partial sums in flight, subtracted in chunk order: one at a time each load was a dependent ~1.5 us round trip past the L2 —
      //  10 chunks x 4 tiles = 60 us of a 69 us launch on the 5-agent map's upper levels)
      if (act)
        for (int q0 = 0; q0 < nch; q0 += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            t[u] = q0 + u < nch ? __hip_atomic_load(scr_f + ((size_t)p * nchunk + q0 + u) * kTile + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) v -= t[u];
        }
      double dv[PB], Lc[7][PB];
#pragma unroll
      for (int r = 0; r < PB; ++r) dv[r] = act ? Dinv[jbc * 256 + r * PB + cl] : 0.0;
#pragma unroll
      for (int jb = 1; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < PB; ++r) Lc[jb - 1][r] = (act && jbc < jb) ? M[(size_t)(k0 + PB * jb + r) * ld + k0 + c] : 0.0;
#pragma unroll
      for (int jb = 7; jb >= 0; --jb) {
        if (act && jbc == jb) sv[c] = v;
        __syncthreads();
        if (act && jbc == jb) {
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int r = 0; r < PB; r += 4) {
            s0 += dv[r] * sv[PB * jb + r]; s1 += dv[r + 1] * sv[PB * jb + r + 1];
            s2 += dv[r + 2] * sv[PB * jb + r + 2]; s3 += dv[r + 3] * sv[PB * jb + r + 3];
          }
          sx[c] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (jb > 0 && act && jbc < jb) {
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int r = 0; r < PB; r += 4) {
            s0 += Lc[jb > 0 ? jb - 1 : 0][r] * sx[PB * jb + r]; s1 += Lc[jb > 0 ? jb - 1 : 0][r + 1] * sx[PB * jb + r + 1];
            s2 += Lc[jb > 0 ? jb - 1 : 0][r + 2] * sx[PB * jb + r + 2]; s3 += Lc[jb > 0 ? jb - 1 : 0][r + 3] * sx[PB * jb + r + 3];
          }
          v -= (s0 + s1) + (s2 + s3);
        }
      }
    }
    // own unknowns of this tile -> solution vector
    {
      const int n = xf.own_dims[node];
      const int* gi = xf.gidx + xf.own_g[node];
      if (tid < kTile && k0 + tid < n) xf.x[gi[k0 + tid]] = sx[tid];
    }
    // y[cols left of the tile] -= L[tile rows, cols]^T x_p: two groups of 64 rows per 128 columns, added in group order
    if (p > 0) {
      __shared__ double part2[2][kTile];
      const int cl = tid & 127, rg = tid >> 7;
      for (int c0 = 0; c0 < k0; c0 += kTile) {
        const double* Lc2 = M + (size_t)(k0 + 64 * rg) * ld + c0 + cl;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
        for (int r = 0; r < 64; r += 4) {
          a0 += Lc2[(size_t)r * ld] * sx[64 * rg + r]; a1 += Lc2[(size_t)(r + 1) * ld] * sx[64 * rg + r + 1];
          a2 += Lc2[(size_t)(r + 2) * ld] * sx[64 * rg + r + 2]; a3 += Lc2[(size_t)(r + 3) * ld] * sx[64 * rg + r + 3];
        }
        part2[rg][cl] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (rg == 0) y[c0 + cl] -= part2[0][cl] + part2[1][cl];
        __syncthreads();
      }
    }
  }
}

// ---- Backward substitution of fronts with MANY interior tiles as ONE launch: a pipeline of workgroups (round 6) -----------------------------------
// The top of the elimination tree holds single fronts of 4 .. 16 interior tiles; one launch per tile (k_bwd_step_sub) costs 8.8 us a tile — a launch
// boundary (3.6 us under load) and two dependent memory latencies around 1 us of arithmetic — 141 us for the 5-agent map's root. Here every interior
// tile p has its own workgroup, all in one launch:
//   helpers     (tile ct, 256-row chunk k of the border): partial sums of y_ct -= L[border rows, ct]^T x_border, as in k_bwd_front
//   pipeline p  v = y_p - (its helpers' sums) - sum_{q > p} L[q, p]^T x_q, taking the x_q in the order they are solved (q = last .. p + 1; tile L[q, p]
//               is in registers before x_q arrives), then x_p = L_pp^-T v with the block inverses, publishes x_p, writes it to the solution vector.
// Hand-over between workgroups: the DATA WORDS are the flags. Every slot of `scr` / `xpub` holds kPipeEmpty (a NaN payload no arithmetic produces)
// between launches; the producer stores its values with agent-scope stores, the consumer polls the word it needs until it differs
// (tools/pipe_probe.hip: 0.6 us per hand-over across XCDs, against 1.15 for flag + data and 3.6 for a launch boundary). A helper's slot has one
// reader, which puts kPipeEmpty back; x_q is read by every pipeline workgroup p < q, the LAST of which (p = 0) restores the front's slots — by
// induction over the chain every other read has completed when workgroup 0 holds x_1. No counters, no clearing launch.
// Forward progress: a workgroup only waits for workgroups with a LOWER linear index (helpers first, then the tiles from the last to the first),
// which the dispatcher has started before it — no co-residency assumption. A wait beyond `limit` raises the gate's dead flag (CholAux::gate_failed:
// the solve is repeated on the launch-per-tile path and the context stays there).
static constexpr unsigned long long kPipeEmpty = 0x7ff8c0f6a11d0e5full;
struct BwdPipeArgs {
  const double* M; int tI, T, nchunk; double* y; const double* Dinv_all; size_t bsL, bsR; const long long* btab; const int* live; BwdXfer xf;
  double *scr, *xpub; int *dead, *dead_h; long long limit; int check;   // check: polls between two looks at the clock, minus one (a power of two)
  int fault;   // dev aid (COVGPU_PIPE_FAULT=1, tests): the first tile of every chain withholds its result — whoever waits for it runs into the limit
  int tree;    // k_bwd_tree (several levels in one launch): the ancestors' unknowns are POLLED in the solution vector (its entries of the merged levels
               // hold kPipeEmpty since k_nd_assemble), a front's own unknowns go there with agent-scope stores
};
COV_DEV double pipe_take(double* slot, const BwdPipeArgs& g, bool restore) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(slot);
  unsigned long long v;
  long long t0 = 0; int spins = 0;
  while ((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kPipeEmpty) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & g.check) == 0) {
      if (t0 == 0) t0 = wall_clock64();
      else if (wall_clock64() - t0 > g.limit || __hip_atomic_load(g.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (atomicExch(g.dead, 1) == 0) { g.dead_h[1] = -2; g.dead_h[2] = (int)blockIdx.x; g.dead_h[3] = (int)blockIdx.y; __threadfence_system(); g.dead_h[0] = 1; }
        v = 0ull; break;
      }
    }
  }
  if (restore) __hip_atomic_store(w, kPipeEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)v);
}
// helper workgroup hx = tile * nchunk + chunk of front `batch`: its kPipeChunk border rows' share of y_tile, into its slot of `scr`.
// ALL its rows of L are in registers (32 double2 a lane) before the ancestors' unknowns are gathered: what follows their arrival is arithmetic only.
// (First form: 256 rows a workgroup in eight batches of eight loads — eight dependent memory latencies, ~10 us of every level's ~20.)
COV_DEV void bwd_pipe_helper(const BwdPipeArgs& g, int hx, int batch, double* sxg /* [kPipeChunk] */, double2 (*part)[64]) {
  const int tid = threadIdx.x;
  const int nIt = g.live[2 * batch];
  const int node = g.xf.first + batch;
  const int nst = g.xf.st_dims[node];
  const int nch = (nst + kPipeChunk - 1) / kPipeChunk;
  const double* M = g.M + (size_t)g.btab[2 * batch];
  const size_t ld = (size_t)g.btab[2 * batch + 1];
  double* scr_f = g.scr + (size_t)batch * g.T * g.nchunk * kTile;
  const int ct = hx / g.nchunk, k = hx % g.nchunk;
  if (ct >= nIt || k >= nch) return;
  const int r0 = g.tI * kTile + kPipeChunk * k, nr = min(kPipeChunk, nst - kPipeChunk * k);
  const int lane = tid & 63, wv = tid >> 6;
  const int col = ct * kTile + 2 * lane;
  const double* Lp = M + (size_t)r0 * ld + col;
  static_assert(kPipeChunk == 128, "32 rows a wave");
  double2 v[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) v[u] = (wv + 4 * u < nr) ? *reinterpret_cast<const double2*>(Lp + (size_t)(wv + 4 * u) * ld) : double2{0.0, 0.0};
  const int* gi = g.xf.gidx + g.xf.st_g[node] + kPipeChunk * k;
  if (tid < kPipeChunk) sxg[tid] = tid < nr ? (g.tree ? pipe_take(g.xf.x + gi[tid], g, false) : g.xf.x[gi[tid]]) : 0.0;
  __syncthreads();
  double2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < 32; u += 2) {
    const double x0 = sxg[wv + 4 * u], x1 = sxg[wv + 4 * u + 4];
    a0.x += v[u].x * x0; a0.y += v[u].y * x0; a1.x += v[u + 1].x * x1; a1.y += v[u + 1].y * x1;
  }
  part[wv][lane] = double2{a0.x + a1.x, a0.y + a1.y};
  __syncthreads();
  if (wv == 0) {
    const double2 a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
    double* dst = scr_f + ((size_t)ct * g.nchunk + k) * kTile + 2 * lane;
    __hip_atomic_store(dst, ((a.x + b.x) + c.x) + d.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, ((a.y + b.y) + c.y) + d.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
COV_DEV void bwd_pipe_body(const BwdPipeArgs& g, const int bx, const int batch) {   // bx: the workgroup's index inside its front (helpers first)
  const int tid = threadIdx.x;
  const int nIt = g.live[2 * batch];            // real interior tiles of this front
  const int node = g.xf.first + batch;
  const int nst = g.xf.st_dims[node];
  const int nch = (nst + kPipeChunk - 1) / kPipeChunk;   // row chunks of this front's border (0: the root)
  const double* M = g.M + (size_t)g.btab[2 * batch];
  const size_t ld = (size_t)g.btab[2 * batch + 1];
  double* scr_f = g.scr + (size_t)batch * g.T * g.nchunk * kTile;   // [tile][chunk][128]
  double* xpub_f = g.xpub + (size_t)batch * g.T * kTile;            // [tile][128]
  const int nhelp = g.T * g.nchunk;
  __shared__ double2 part[4][64];
  __shared__ double sx[kTile], sv[kTile], sxq[2][kTile], part2[2][kTile];
  if (bx < nhelp) { bwd_pipe_helper(g, bx, batch, &sxq[0][0], part); return; }
  const int p = g.T - 1 - (bx - nhelp);
  if (p >= nIt) return;
  const int k0 = p * kTile;
  double* y = g.y + (size_t)batch * g.bsR;
  const double* Dinv = g.Dinv_all + (size_t)batch * g.bsL + (size_t)p * kTile * kTile;
  const int c = tid & 127, h = tid >> 7, jbc = c >> 4, cl = c & 15;
  const bool act = tid < kTile;
  // ---- everything this tile will need that is already there: the first tile of its column, y_p, the diagonal tile
  double v = act ? y[k0 + c] : 0.0;
  const int n_own = g.xf.own_dims[node];
  const int gi_own = (act && k0 + c < n_own) ? g.xf.gidx[g.xf.own_g[node] + k0 + c] : -1;   // (loaded here, not between the last sum and its store)
  double dv[PB];
#pragma unroll
  for (int r = 0; r < PB; ++r) dv[r] = act ? Dinv[jbc * 256 + r * PB + cl] : 0.0;
  // (the diagonal tile's blocks below the block diagonal go to LDS, packed by block row jb = 1 .. 7: [16 rows][16 jb columns] at 128 jb (jb - 1) —
  //  in registers, as k_bwd_step_sub holds them, they and the column tile above exceed the register file)
  extern __shared__ double sLc[];
#pragma unroll
  for (int jb = 1; jb < 8; ++jb)
#pragma unroll
    for (int rr = 0; rr < PB; rr += 2) {
      const int r = rr + h;
      if (c < PB * jb) sLc[128 * jb * (jb - 1) + r * PB * jb + c] = M[(size_t)(k0 + PB * jb + r) * ld + k0 + c];
    }
  // ---- the border's share (helpers), in chunk order
  if (act)
    for (int q0 = 0; q0 < nch; q0 += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = q0 + u < nch ? pipe_take(scr_f + ((size_t)p * g.nchunk + q0 + u) * kTile + c, g, true) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) v -= t[u];
    }
  // ---- the tiles solved before this one, in the order they are solved
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#ifdef COVGPU_PIPE_PROBE
  long long tp0 = wall_clock64(), tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
#endif
  for (int q = nIt - 1; q > p; --q) {   // (tile L[q, p] is loaded at the top of its step, before the poll: see k_bwd_pipe64)
    double Lq[64];
    {
      const double* src = M + (size_t)(q * kTile + 64 * h) * ld + k0 + c;
#pragma unroll
      for (int r = 0; r < 64; ++r) Lq[r] = src[(size_t)r * ld];
    }
    if (act) sxq[q & 1][c] = pipe_take(xpub_f + (size_t)q * kTile + c, g, false);
    lds_barrier();
#ifdef COVGPU_PIPE_PROBE
    tp1 = wall_clock64();
#endif
    const double* xs = &sxq[q & 1][64 * h];
#pragma unroll
    for (int r = 0; r < 64; r += 4) { a0 += Lq[r] * xs[r]; a1 += Lq[r + 1] * xs[r + 1]; a2 += Lq[r + 2] * xs[r + 2]; a3 += Lq[r + 3] * xs[r + 3]; }
  }
  part2[h][c] = (a0 + a1) + (a2 + a3);
  lds_barrier();
  if (act) v -= part2[0][c] + part2[1][c];
#ifdef COVGPU_PIPE_PROBE
  tp2 = wall_clock64();
#endif
  // ---- x_p = L_pp^-T v (block inverses; the same steps as k_bwd_step_sub)
#pragma unroll
  for (int jb = 7; jb >= 0; --jb) {
    if (act && jbc == jb) sv[c] = v;
    lds_barrier();
    if (act && jbc == jb) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int r = 0; r < PB; r += 4) {
        s0 += dv[r] * sv[PB * jb + r]; s1 += dv[r + 1] * sv[PB * jb + r + 1];
        s2 += dv[r + 2] * sv[PB * jb + r + 2]; s3 += dv[r + 3] * sv[PB * jb + r + 3];
      }
      sx[c] = (s0 + s1) + (s2 + s3);
    }
    lds_barrier();
    if (jb > 0 && act && jbc < jb) {
      const double* Lt = sLc + 128 * jb * (jb - 1) + c;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int r = 0; r < PB; r += 4) {
        s0 += Lt[r * PB * jb] * sx[PB * jb + r]; s1 += Lt[(r + 1) * PB * jb] * sx[PB * jb + r + 1];
        s2 += Lt[(r + 2) * PB * jb] * sx[PB * jb + r + 2]; s3 += Lt[(r + 3) * PB * jb] * sx[PB * jb + r + 3];
      }
      v -= (s0 + s1) + (s2 + s3);
    }
  }
#ifdef COVGPU_PIPE_PROBE
  tp3 = wall_clock64();
#endif
  if (act && p > 0 && !(g.fault && p == nIt - 1)) __hip_atomic_store(xpub_f + (size_t)p * kTile + c, sx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef COVGPU_PIPE_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tp4 = wall_clock64();
  if (tid == 0 && nst == 0) printf("pipe tile %2d: start %lld  last x in +%lld  gemv+reduce +%lld  solve +%lld  published +%lld (10 ns ticks)\n", p, tp0 % 100000000ll, tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
#endif
  if (gi_own >= 0) {   // own unknowns of this tile -> solution vector
    if (g.tree) __hip_atomic_store(g.xf.x + gi_own, sx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else g.xf.x[gi_own] = sx[c];
  }
  // the last tile of the chain has seen every x_q of the front, and so has everybody else by then: the slots are free again
  if (p == 0)
    for (int i = kTile + tid; i < nIt * kTile; i += 256)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(xpub_f + i), kPipeEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_bwd_pipe(BwdPipeArgs g) { bwd_pipe_body(g, (int)blockIdx.x, (int)blockIdx.y); }
// SEVERAL LEVELS of the tree in one launch (the bottom levels: hundreds of fronts of one or two interior tiles, ~20 us a level as launches of their
// own — a boundary, the gather of the ancestors' unknowns, the rows of L, the sums, the tile, the scatter, each a dependent memory latency). The
// levels stand in the grid from the highest to the lowest; a front's helpers POLL its ancestors' unknowns in the solution vector, whose entries of the
// merged levels k_nd_assemble filled with kPipeEmpty at the start of the solve (entries of higher levels hold values long since: the poll falls
// through), and the tile workgroups publish a front's own unknowns there with agent-scope stores. A workgroup still only waits for workgroups with a
// lower linear index. What a level loads that does not depend on the levels above — its rows of L, its block inverses — is in flight while it waits.
struct BwdTreeArgs { BwdPipeArgs base; BwdTreeLevel lev[kBwdTreeMax]; int wg0[kBwdTreeMax + 1]; int nlev; };
__global__ __launch_bounds__(256) void k_bwd_tree(BwdTreeArgs a) {
  int l = 0;
  while (l + 1 < a.nlev && (int)blockIdx.x >= a.wg0[l + 1]) ++l;
  const BwdTreeLevel& L = a.lev[l];
  BwdPipeArgs g = a.base;
  g.tI = L.tI; g.T = L.T; g.nchunk = L.nchunk; g.y = L.y; g.Dinv_all = L.Dinv; g.bsL = L.bsL; g.bsR = L.bsR; g.btab = L.btab; g.live = L.live; g.xf.first = L.first;
  g.scr = a.base.scr + L.scr_off; g.xpub = a.base.scr + L.xpub_off; g.tree = 1;
  const int per = L.T * L.nchunk + L.T, local = (int)blockIdx.x - a.wg0[l];
  bwd_pipe_body(g, local % per, local / per);
}
// The same pipeline for the FEW fronts at the top of the tree (at most 128 interior tiles in the launch), where the chain of tiles is the whole cost:
// measured on the 5-agent map's root, a tile of k_bwd_pipe costs 4.4 us — 0.6 hand-over, 1.0 the product with the newest x_q, 2.8 the eight-step
// substitution with the 16x16 block inverses (two barriers a step). Here every tile's workgroup first forms the INVERSES OF THE TWO 64x64 DIAGONAL
// BLOCKS of its tile by two doubling steps from the 16x16 inverses ([[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]; ~2 us, while the chain is
// still far away for all but the first tile), and the substitution becomes three products of 16 terms a lane with wave-local sums:
//   x_hi = C^-T v_hi;  x_lo = A^-T (v_lo - B^T x_hi)        (three barriers instead of sixteen)
// and the product with x_q sums inside a wave too (column c's two half sums sit 32 lanes apart). The order of the workgroups is k_bwd_pipe's (helpers
// first): a workgroup takes a whole CU (111 KB of LDS), and with the tile workgroups first two contexts running this launch at once (virtual ranks of a
// sharded solve on one GPU) could fill every CU with tile workgroups that wait for helpers which find no room. Launches of more than 128 tile workgroups
// use k_bwd_pipe (eight to a CU).
constexpr int kP64W = 0, kP64S1 = 2 * 64 * 65, kP64S2 = kP64S1 + 4 * 16 * 17, kP64Y = kP64S2 + 2 * 32 * 33, kP64V = kP64Y + 2 * 32 * 33, kP64Doubles = kP64V + 128 + 128 + 64 + 256;
COV_DEV void bwd_pipe64_body(const BwdPipeArgs& g, const int bx, const int batch) {
  extern __shared__ __attribute__((aligned(16))) double sm64[];
  const int tid = threadIdx.x;
  const int nhelp = g.T * g.nchunk;
  if (bx < nhelp) { bwd_pipe_helper(g, bx, batch, sm64, reinterpret_cast<double2(*)[64]>(sm64 + 256)); return; }
  const int nIt = g.live[2 * batch];
  const int p = g.T - 1 - (bx - nhelp);
  if (p >= nIt) return;
  const int node = g.xf.first + batch;
  const int nch = (g.xf.st_dims[node] + kPipeChunk - 1) / kPipeChunk;
  const double* M = g.M + (size_t)g.btab[2 * batch];
  const size_t ld = (size_t)g.btab[2 * batch + 1];
  double* scr_f = g.scr + (size_t)batch * g.T * g.nchunk * kTile;
  double* xpub_f = g.xpub + (size_t)batch * g.T * kTile;
  const int k0 = p * kTile;
  double* y = g.y + (size_t)batch * g.bsR;
  const double* Dinv = g.Dinv_all + (size_t)batch * g.bsL + (size_t)p * kTile * kTile;
  double (*W)[64][65] = reinterpret_cast<double (*)[64][65]>(sm64 + kP64W);     // the two 64x64 inverses (lower; zeros above the diagonal)
  double (*S1)[16][17] = reinterpret_cast<double (*)[16][17]>(sm64 + kP64S1);  // L blocks (2t+1, 2t) of the tile, t < 4
  double (*S2)[32][33] = reinterpret_cast<double (*)[32][33]>(sm64 + kP64S2);  // L blocks rows 64u+32.., cols 64u.., u < 2
  double (*Y2)[32][33] = reinterpret_cast<double (*)[32][33]>(sm64 + kP64Y);
  double (*Y1)[16][17] = reinterpret_cast<double (*)[16][17]>(sm64 + kP64Y);
  double* sx = sm64 + kP64V; double* sv = sx + 128; double* svlo = sv + 128; double* sxq = svlo + 64;   // sxq: [2][128]
  const int w = tid >> 6, l = tid & 63;
  const int c = 32 * w + (l & 31), h = l >> 5;      // product with x_q: column c, row half h
  const bool vh = h == 0;                           // ... and the lane that holds v[c]
  const int cc = 16 * w + (l & 15), gq = l >> 4;    // substitution: column cc of a 64-block, row group gq
  // ---- loads of everything that is already there: what the inverses need first (loads return in order), then the first tile of the column, y_p, B.
  //      (Every barrier below orders LDS traffic only: __syncthreads() would also wait for the loads and stores in flight — a memory latency each)
  double dreg[8], s1reg[4], s2reg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dreg[i] = Dinv[tid + 256 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i, t = e >> 8, rr = (e >> 4) & 15, cx = e & 15; s1reg[i] = M[(size_t)(k0 + 32 * t + 16 + rr) * ld + k0 + 32 * t + cx]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int e = tid + 256 * i, u = e >> 10, rr = (e >> 5) & 31, cx = e & 31; s2reg[i] = M[(size_t)(k0 + 64 * u + 32 + rr) * ld + k0 + 64 * u + cx]; }
  double v = vh ? y[k0 + c] : 0.0;
  // (where this tile's unknowns go in the solution vector: the index loads would sit between the sums and the stores of the last steps)
  const int n_own = g.xf.own_dims[node];
  const int* gi = g.xf.gidx + g.xf.own_g[node];
  const int gi_hi = (gq == 0 && k0 + 64 + cc < n_own) ? gi[k0 + 64 + cc] : -1, gi_lo = (gq == 0 && k0 + cc < n_own) ? gi[k0 + cc] : -1;
  {
    for (int i = tid; i < 2 * 64 * 65; i += 256) sm64[kP64W + i] = 0.0;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int e = tid + 256 * i, blk = e >> 8, rr = (e >> 4) & 15, cx = e & 15; W[blk >> 2][16 * (blk & 3) + rr][16 * (blk & 3) + cx] = dreg[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; S1[e >> 8][(e >> 4) & 15][e & 15] = s1reg[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int e = tid + 256 * i; S2[e >> 10][(e >> 5) & 31][e & 31] = s2reg[i]; }
    lds_barrier();
  }
  double bb[16];   // (B of this tile: needed last, loaded once the staging registers are free)
#pragma unroll
  for (int i = 0; i < 16; ++i) bb[i] = M[(size_t)(k0 + 64 + 16 * gq + i) * ld + k0 + cc];
  {  // ---- 16 -> 32: four pairs, one per wave
    const int t = w, u = t >> 1, o = 32 * (t & 1), c1 = l & 15, r0 = 4 * (l >> 4);
    double yv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double d = W[u][o + k][o + c1];
#pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] += S1[t][r0 + i][k] * d;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) Y1[t][r0 + i][c1] = yv[i];
    lds_barrier();
    double xv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double yk = Y1[t][k][c1];
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] -= W[u][o + 16 + r0 + i][o + 16 + k] * yk;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) W[u][o + 16 + r0 + i][o + c1] = xv[i];
    lds_barrier();
  }
  {  // ---- 32 -> 64: two pairs, two waves each
    const int u = tid >> 7, l2 = tid & 127, c2 = l2 & 31, r0 = 8 * (l2 >> 5);
    double yv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double d = W[u][k][c2];
#pragma unroll
      for (int i = 0; i < 8; ++i) yv[i] += S2[u][r0 + i][k] * d;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) Y2[u][r0 + i][c2] = yv[i];
    lds_barrier();
    double xv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double yk = Y2[u][k][c2];
#pragma unroll
      for (int i = 0; i < 8; ++i) xv[i] -= W[u][32 + r0 + i][32 + k] * yk;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) W[u][32 + r0 + i][c2] = xv[i];
    lds_barrier();
  }
  double w1[16], w0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { w1[i] = W[1][16 * gq + i][cc]; w0[i] = W[0][16 * gq + i][cc]; }
  // ---- the border's share (helpers), in chunk order
  if (vh)
    for (int q0 = 0; q0 < nch; q0 += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = q0 + u < nch ? pipe_take(scr_f + ((size_t)p * g.nchunk + q0 + u) * kTile + c, g, true) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) v -= t[u];
    }
  // ---- the tiles solved before this one, in the order they are solved
  // (tile L[q, p] is loaded at the top of the step that uses it, before x_q is polled — loads return in order, so the poll comes back behind it;
  //  a workgroup that keeps up with the chain spends that latency waiting for x_q anyway. Prefetching it a step ahead into the same registers made
  //  the compiler rotate 128 registers through copies every step: 1 us of a 3.8 us step)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#ifdef COVGPU_PIPE_PROBE
  long long tp0 = wall_clock64(), tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
#endif
  for (int q = nIt - 1; q > p; --q) {
    double Lq[64];
    {
      const double* src = M + (size_t)(q * kTile + 64 * h) * ld + k0 + c;
#pragma unroll
      for (int r = 0; r < 64; ++r) Lq[r] = src[(size_t)r * ld];
    }
    if (vh) sxq[128 * (q & 1) + c] = pipe_take(xpub_f + (size_t)q * kTile + c, g, false);
    lds_barrier();
#ifdef COVGPU_PIPE_PROBE
    tp1 = wall_clock64();
#endif
    const double2* xs = reinterpret_cast<const double2*>(sxq + 128 * (q & 1) + 64 * h);
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
      const double2 x0 = xs[r], x1 = xs[r + 1];
      a0 += Lq[2 * r] * x0.x; a1 += Lq[2 * r + 1] * x0.y; a2 += Lq[2 * r + 2] * x1.x; a3 += Lq[2 * r + 3] * x1.y;
    }
  }
  {
    double acc = (a0 + a1) + (a2 + a3);
    acc += __shfl_xor(acc, 32, 64);
    if (vh) { v -= acc; sv[c] = v; }
  }
  lds_barrier();
#ifdef COVGPU_PIPE_PROBE
  tp2 = wall_clock64();
#endif
  {  // x_hi = C^-T v_hi
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) { s0 += w1[i] * sv[64 + 16 * gq + i]; s1 += w1[i + 1] * sv[64 + 16 * gq + i + 1]; }
    double sum = s0 + s1;
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    if (gq == 0) {
      sx[64 + cc] = sum;
      if (p > 0 && !(g.fault && p == nIt - 1)) __hip_atomic_store(xpub_f + (size_t)p * kTile + 64 + cc, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gi_hi >= 0) { if (g.tree) __hip_atomic_store(g.xf.x + gi_hi, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else g.xf.x[gi_hi] = sum; }
    }
  }
  lds_barrier();
  {  // v_lo - B^T x_hi
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) { s0 += bb[i] * sx[64 + 16 * gq + i]; s1 += bb[i + 1] * sx[64 + 16 * gq + i + 1]; }
    double sum = s0 + s1;
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    if (gq == 0) svlo[cc] = sv[cc] - sum;
  }
  lds_barrier();
  {  // x_lo = A^-T (...)
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) { s0 += w0[i] * svlo[16 * gq + i]; s1 += w0[i + 1] * svlo[16 * gq + i + 1]; }
    double sum = s0 + s1;
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    if (gq == 0) {
      if (p > 0 && !(g.fault && p == nIt - 1)) __hip_atomic_store(xpub_f + (size_t)p * kTile + cc, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gi_lo >= 0) { if (g.tree) __hip_atomic_store(g.xf.x + gi_lo, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else g.xf.x[gi_lo] = sum; }
    }
  }
#ifdef COVGPU_PIPE_PROBE
  tp3 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tp4 = wall_clock64();
  if (tid == 0 && nch == 0) printf("pipe64 tile %2d: start %lld  last x in +%lld  product +%lld  solve +%lld  stores done +%lld (10 ns ticks)\n", p, tp0 % 100000000ll, tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
#endif
  if (p == 0)
    for (int i = kTile + tid; i < nIt * kTile; i += 256)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(xpub_f + i), kPipeEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_bwd_pipe64(BwdPipeArgs g) { bwd_pipe64_body(g, (int)blockIdx.x, (int)blockIdx.y); }
// ... and the TOP levels of the tree in one launch of this form (at most 128 tile workgroups in all): k_bwd_tree's scheme
__global__ __launch_bounds__(256) void k_bwd_tree64(BwdTreeArgs a) {
  int l = 0;
  while (l + 1 < a.nlev && (int)blockIdx.x >= a.wg0[l + 1]) ++l;
  const BwdTreeLevel& L = a.lev[l];
  BwdPipeArgs g = a.base;
  g.tI = L.tI; g.T = L.T; g.nchunk = L.nchunk; g.y = L.y; g.Dinv_all = L.Dinv; g.bsL = L.bsL; g.bsR = L.bsR; g.btab = L.btab; g.live = L.live; g.xf.first = L.first;
  g.scr = a.base.scr + L.scr_off; g.xpub = a.base.scr + L.xpub_off; g.tree = 1;
  const int per = L.T * L.nchunk + L.T, local = (int)blockIdx.x - a.wg0[l];
  bwd_pipe64_body(g, local % per, local / per);
}
__global__ void k_pipe_fill(unsigned long long* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = kPipeEmpty;
}
unsigned long long pipe_empty_word() { return kPipeEmpty; }
void launch_pipe_fill(double* buf, size_t n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_pipe_fill, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(buf), n);
}
void launch_bwd_tree(const double* M, const BwdTreeLevel* lev, int nlev, BwdXfer xf, double* pipe, int* dead, int* dead_h, double timeout_s, hipStream_t st, bool form64) {
  BwdTreeArgs a;
  a.base = BwdPipeArgs{M, 0, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, xf, pipe, pipe, dead, dead_h, (long long)(timeout_s * 1e8), 2047, 0, 1};
  static const int check = getenv("COVGPU_PIPE_SPIN_CHECK") ? std::max(1, atoi(getenv("COVGPU_PIPE_SPIN_CHECK"))) : 2048;
  static const int fault = getenv("COVGPU_PIPE_FAULT") ? atoi(getenv("COVGPU_PIPE_FAULT")) : 0;
  a.base.check = check - 1; a.base.fault = fault;
  a.nlev = nlev; a.wg0[0] = 0;
  size_t off = 0;
  for (int l = 0; l < nlev; ++l) {
    a.lev[l] = lev[l];
    a.lev[l].scr_off = off; off += (size_t)lev[l].nbt * lev[l].T * lev[l].nchunk * kTile;
    a.lev[l].xpub_off = off; off += (size_t)lev[l].nbt * lev[l].T * kTile;
    a.wg0[l + 1] = a.wg0[l] + lev[l].nbt * (lev[l].T * lev[l].nchunk + lev[l].T);
  }
  constexpr size_t lds = (size_t)128 * 7 * 8 * sizeof(double), lds64 = (size_t)kP64Doubles * sizeof(double);
  static std::atomic<unsigned long long> seen{0};
  if (first_use_on_device(seen)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_tree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_tree64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
  }
  if (form64) hipLaunchKernelGGL(k_bwd_tree64, dim3(a.wg0[nlev]), dim3(256), lds64, st, a);
  else hipLaunchKernelGGL(k_bwd_tree, dim3(a.wg0[nlev]), dim3(256), lds, st, a);
}
void launch_bwd_pipe(const double* S, int tI, int ntiles, int nchunk, double* y, const double* Linv, int nbt, size_t sL, size_t sR, hipStream_t st,
                     const long long* btab, const int* live, BwdXfer xf, double* pipe, int* dead, int* dead_h, double timeout_s) {
  BwdPipeArgs g{S, tI, ntiles, nchunk, y, Linv, sL, sR, btab, live, xf, pipe, pipe + (size_t)nbt * ntiles * nchunk * kTile, dead, dead_h, (long long)(timeout_s * 1e8), 2047, 0, 0};
  static const int fault = getenv("COVGPU_PIPE_FAULT") ? atoi(getenv("COVGPU_PIPE_FAULT")) : 0;
  g.fault = fault;
  static const int check = getenv("COVGPU_PIPE_SPIN_CHECK") ? std::max(1, atoi(getenv("COVGPU_PIPE_SPIN_CHECK"))) : 2048;   // (the test of the fallback: 1)
  g.check = check - 1;
  constexpr size_t lds = (size_t)128 * 7 * 8 * sizeof(double);   // the packed blocks of the diagonal tile
  static std::atomic<unsigned long long> seen{0}, seen64{0};
  if (first_use_on_device(seen)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // few tiles in the launch: the form with the 64x64 inverses (a whole CU per tile workgroup)
  static const int max64 = getenv("COVGPU_BWD_PIPE64") ? atoi(getenv("COVGPU_BWD_PIPE64")) : 128;
  if (nbt * ntiles <= max64) {
    constexpr size_t lds64 = (size_t)kP64Doubles * sizeof(double);
    if (first_use_on_device(seen64)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_pipe64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
    hipLaunchKernelGGL(k_bwd_pipe64, dim3(ntiles * nchunk + ntiles, nbt), dim3(256), lds64, st, g);
    return;
  }
  hipLaunchKernelGGL(k_bwd_pipe, dim3(ntiles * nchunk + ntiles, nbt), dim3(256), lds, st, g);
}

// ---- launch wrappers (k_chol.hip schedules them) ----------------------------------------------------------------------
void launch_bwd_given(const double* S, size_t ld, int r0, int r1, double* y, double* x, int ncol, int nbt, size_t sM, size_t sR, hipStream_t st,
                      const long long* btab, const int* live, int tI, BwdXfer xf) {
  if (r1 <= r0 || ncol <= 0) return;
  hipLaunchKernelGGL(k_bwd_given, dim3((ncol + 127) / 128, nbt), dim3(256), (size_t)(r1 - r0) * sizeof(double), st, S, ld, r0, r1, y, (const double*)x, ncol, sM,
                     sR, btab, live, tI, xf);
}

bool launch_potrf_panel(double* S, size_t ld, int t0, int w, double* Linv, int* flag, double* b, int npad, int nbt, size_t sM, size_t sL, size_t sR,
                        hipStream_t st, const long long* btab, int nb, const int* own, const int* list, int n_big, int n_small, DevSignal sa, DevSignal sb) {
  if (nb < 0) nb = 8 * w;
  if (nb == 0) return false;  // an all-padding panel of every front of the batch: L = I, Dinv = I, y = 0 are in place
  static std::atomic<unsigned long long> seen{0};
  if (first_use_on_device(seen)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_panel4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelLds4);
  }
  double* Lp = Linv + (size_t)t0 * kTile * kTile;
  const double* yb = b ? b + npad : nullptr;
  if (list == nullptr) {   // every front of the batch in the sixteen-wave form (arrow blocks of the pose graph, the dense solve)
    hipLaunchKernelGGL(k_potrf_panel, dim3(nbt), dim3(64 * NW), kPanelLds, st, S, ld, t0 * kTile, nb, Lp, flag, (const double*)b, (double*)yb, sM, sL, sR, btab, own, list, sa, sb);
    return true;
  }
  // list[0 .. n_big): fronts with more than 128 real columns in this panel | list[n_big .. n_big + n_small): the others that have any (four waves)
  // The four-wave form runs three fronts per CU on the same three matrix pipes: per front it is slower (one tile wave per SIMD instead of four), so it
  // only pays when the small fronts outnumber the CUs — 5-agent map, 182 speed-bias segments: 245 it/s with it against 250 without.
  static const int small_min = getenv("COVGPU_POTRF4_MIN") ? atoi(getenv("COVGPU_POTRF4_MIN")) : 384;
  if (n_small <= small_min) { n_big += n_small; n_small = 0; }
  // (the pending records go out with the first launch that happens; false: nothing was launched — the caller publishes them itself)
  if (n_big > 0) hipLaunchKernelGGL(k_potrf_panel, dim3(n_big), dim3(64 * NW), kPanelLds, st, S, ld, t0 * kTile, nb, Lp, flag, (const double*)b, (double*)yb, sM, sL, sR, btab, own, list, sa, sb);
  if (n_small > 0)
    hipLaunchKernelGGL(k_potrf_panel4, dim3(n_small), dim3(256), kPanelLds4, st, S, ld, t0 * kTile, std::min(nb, 8), Lp, flag, (const double*)b, (double*)yb, sM, sL, sR, btab, own,
                       list + n_big, n_big > 0 ? DevSignal() : sa, n_big > 0 ? DevSignal() : sb);
  return n_big > 0 || n_small > 0;
}

void launch_trsm_sub(double* S, size_t ld, int t0, int w, int r0, int r1, const double* Linv, double* b, int npad, int nbt, size_t sM, size_t sL,
                     size_t sR, const int* live, int tI, hipStream_t st, bool chain, const long long* btab, int nb, const int* own) {
  if (r1 <= r0 || nb == 0) return;
  TrsmSubArgs g{S, ld, t0 * kTile, r0 * kTile, Linv + (size_t)t0 * kTile * kTile, b, b ? b + npad : nullptr, sM, sL, sR, live, tI, chain ? 1 : 0, btab, own};
  const dim3 grid((r1 - r0) * (kTile / PB), nbt);
  // nb: 16-column blocks of the panel that hold real columns (the rest is identity padding with zeros below: X = A there)
  const int need = nb > 0 ? std::min(nb, 8 * w) : 8 * w;
  static const int four = getenv("COVGPU_TRSM4") == nullptr ? 1 : atoi(getenv("COVGPU_TRSM4"));   // 0: one wave per slab everywhere; 2: four waves on the serial chain only
  if (four == 1 || (four == 2 && chain)) {
    if (need <= 4) hipLaunchKernelGGL(k_trsm_sub4<4>, grid, dim3(256), 0, st, g);
    else if (need <= 8) hipLaunchKernelGGL(k_trsm_sub4<8>, grid, dim3(256), 0, st, g);
    else if (need <= 12) hipLaunchKernelGGL(k_trsm_sub4<12>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(k_trsm_sub4<16>, grid, dim3(256), 0, st, g);
    return;
  }
  if (need <= 4) hipLaunchKernelGGL(k_trsm_sub<4>, grid, dim3(64), 0, st, g);
  else if (need <= 8) hipLaunchKernelGGL(k_trsm_sub<8>, grid, dim3(64), 0, st, g);
  else if (need <= 12) hipLaunchKernelGGL(k_trsm_sub<12>, grid, dim3(64), 0, st, g);
  else hipLaunchKernelGGL(k_trsm_sub<16>, grid, dim3(64), 0, st, g);
}

void launch_bwd_front(const double* S, int tI, int ntiles, int nchunk, double* y, const double* Linv, int nbt, size_t sL, size_t sR, hipStream_t st,
                      const long long* btab, const int* live, BwdXfer xf, int* cnt, double* scr) {
  hipLaunchKernelGGL(k_bwd_front, dim3(ntiles * nchunk, nbt), dim3(256), 0, st, S, tI, nchunk, y, Linv, sL, sR, btab, live, xf, cnt, scr);
}

void launch_bwd_step_sub(const double* S, size_t ld, int p, const double* Linv_p, double* y, double* x, int ncol, int nblocks, int nbt, size_t sM,
                         size_t sL, size_t sR, hipStream_t st, const long long* btab, const int* live, int tI, BwdXfer xf) {
  hipLaunchKernelGGL(k_bwd_step_sub, dim3(nblocks, nbt), dim3(256), 0, st, S, ld, p, Linv_p, y, x, ncol, sM, sL, sR, btab, live, tI, xf);
}

}  // namespace covgpu
