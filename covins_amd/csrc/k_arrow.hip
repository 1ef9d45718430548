// k_arrow.hip — block-arrow solve of the GBA pose system (the linear solve of ceres::Solve(SPARSE_SCHUR),
// optimization_be.cpp:560-567), and its host-side plan.
//
// After the speed-bias chains are eliminated (k_struct.hip) every agent's pose block is dense (C' = C - Y^T Y couples
// all keyframes of one IMU chain), but DIFFERENT agents are coupled only where the map was fused: landmarks observed by
// two agents around a loop closure (placerec_be.cpp:222-285) and the loop edges themselves (optimization_be.cpp:538-556).
// Those cross-agent links touch few keyframes. A vertex cover of them — for every cross link at least one endpoint —
// is the BORDER (the "shared poses" of BASELINE.json's north star); what is left of each agent is an independent block:
//
//        [ D_0            B_0 ]     D_a  dense, order 6 x #interior keyframes of agent a
//   C' = [      D_1       B_1 ]     B_a  couples the block to ITS OWN part of the border only (its agent's border
//        [          ...   ... ]          keyframes + foreign border keyframes that see its landmarks)
//        [ B_0^T B_1^T ... C_b ]    C_b  border system, order 6 x #border keyframes
//
// The linearisation kernels write C' straight into per-block arrow buffers [D_a B_a; B_a^T 0] and into C_b (c_entry in
// common.hpp): no dense 6K x 6K matrix exists. All blocks are eliminated by ONE batched partial MFMA Cholesky (k_chol.hip,
// the same launches as for a single block); their Schur contributions -X_a X_a^T are summed onto C_b in block order,
// C_b is factorised densely, and each block finishes with its own backward substitution. Same arithmetic as the dense
// solve up to the elimination order — the order a fill-reducing sparse Cholesky (the reference's CHOLMOD) would find.
// On the 5-agent map: 40 serial tile steps instead of 103 and ~20x fewer flops. This is also the multi-GPU split:
// blocks are owned by ranks, C_b is the all-reduced part.
#include <algorithm>
#include <queue>
#include <vector>

#include "common.hpp"

namespace covgpu {

// blocks and own-border lists for a given border / ownership (by chain position). Border indices follow the IR keyframe
// order pos_kf[] — the same numbering on every rank of a sharded solve, whatever the rank-local chain positions are.
void gba_plan_build(int K, int nchains, const int* chain_ptr, int npairs, const int* pair_i, const int* pair_j, int nepairs,
                    const int* epair_i, const int* epair_j, const char* border, const char* owned, const int* pos_kf, ArrowHostPlan& out) {
  out = ArrowHostPlan();
  out.blk.assign(K, -2); out.loc.assign(K, 0);
  std::vector<int> chain_of(K), block_of_chain(nchains, -1);
  for (int c = 0; c < nchains; ++c) {
    int n = 0;
    for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) { chain_of[q] = c; if (!border[q] && owned[q]) ++n; }
    if (n == 0) continue;
    block_of_chain[c] = out.nblk++;
    out.nint.push_back(n);
    out.max_int = std::max(out.max_int, n);
    int l = 0;
    for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) if (!border[q] && owned[q]) { out.blk[q] = block_of_chain[c]; out.loc[q] = l++; }
  }
  {  // border index = rank of the keyframe's IR index among the border keyframes
    std::vector<std::pair<int, int>> bk;  // (IR index, position)
    for (int q = 0; q < K; ++q) if (border[q]) bk.push_back({pos_kf[q], q});
    std::sort(bk.begin(), bk.end());
    out.nbk = (int)bk.size();
    out.bpos.resize(out.nbk);
    for (int b = 0; b < out.nbk; ++b) { out.bpos[b] = bk[b].second; out.blk[bk[b].second] = -1; out.loc[bk[b].second] = b; }
  }
  // own border of a block: its chain's border keyframes (Y^T Y couples a whole chain) + border keyframes linked to its interior
  std::vector<std::vector<char>> mark(out.nblk, std::vector<char>(std::max(out.nbk, 1), 0));
  for (int b = 0; b < out.nbk; ++b) { const int a = block_of_chain[chain_of[out.bpos[b]]]; if (a >= 0) mark[a][b] = 1; }
  auto own_link = [&](int x, int y) {  // any structural pair, same chain or not
    if ((out.blk[x] == -1) == (out.blk[y] == -1)) return;
    const int in = out.blk[x] == -1 ? y : x, bd = out.blk[x] == -1 ? x : y;
    if (out.blk[in] >= 0) mark[out.blk[in]][out.loc[bd]] = 1;
  };
  for (int p = 0; p < npairs; ++p) own_link(pair_i[p], pair_j[p]);
  for (int p = 0; p < nepairs; ++p) own_link(epair_i[p], epair_j[p]);
  out.own.assign(out.nblk, {});
  for (int a = 0; a < out.nblk; ++a) {
    for (int b = 0; b < out.nbk; ++b) if (mark[a][b]) out.own[a].push_back(b);
    out.max_own = std::max(out.max_own, (int)out.own[a].size());
  }
}

bool gba_plan_analyse(int K, int nchains, const int* chain_ptr, int npairs, const int* pair_i, const int* pair_j, int nepairs,
                      const int* epair_i, const int* epair_j, bool force, const int* pos_kf, ArrowHostPlan& out) {
  out = ArrowHostPlan();
  if (nchains < 2) return false;
  if (!force && 6 * K < 16 * kTile) return false;  // a handful of panels: the dense solve is already latency-bound
  std::vector<int> chain_of(K);
  for (int c = 0; c < nchains; ++c) for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) chain_of[q] = c;
  // cross-chain links
  std::vector<std::vector<int>> adj(K);
  auto link = [&](int a, int b) { if (chain_of[a] != chain_of[b]) { adj[a].push_back(b); adj[b].push_back(a); } };
  for (int p = 0; p < npairs; ++p) link(pair_i[p], pair_j[p]);
  for (int p = 0; p < nepairs; ++p) link(epair_i[p], epair_j[p]);
  for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  // greedy vertex cover: highest remaining cross degree first (ties: lowest IR index) — deterministic
  std::vector<int> deg(K);
  std::vector<char> border(K, 0), owned(K, 1);
  std::priority_queue<std::pair<int, int>> heap;  // (degree, -IR index)
  std::vector<int> pos_of(K);
  for (int q = 0; q < K; ++q) pos_of[pos_kf[q]] = q;
  for (int k = 0; k < K; ++k) { deg[k] = (int)adj[k].size(); if (deg[k]) heap.push({deg[k], -pos_kf[k]}); }
  while (!heap.empty()) {
    const auto top = heap.top(); heap.pop();
    const int k = pos_of[-top.second];
    if (border[k] || top.first != deg[k]) continue;  // stale entry
    if (deg[k] == 0) continue;
    border[k] = 1;
    for (int v : adj[k]) if (!border[v] && deg[v] > 0) { --deg[v]; if (deg[v]) heap.push({deg[v], -pos_kf[v]}); }
    deg[k] = 0;
  }
  gba_plan_build(K, nchains, chain_ptr, npairs, pair_i, pair_j, nepairs, epair_i, epair_j, border.data(), owned.data(), pos_kf, out);
  if (out.nblk < 2) return false;
  if (force) return true;
  // does it pay? serial tile steps: largest interior + border, against the dense chain of all 6K rows
  const int t_arrow = (6 * out.max_int + kTile - 1) / kTile + (6 * out.nbk + kTile - 1) / kTile, t_dense = (6 * K + kTile - 1) / kTile;
  return 5 * t_arrow <= 4 * t_dense;
}

// identity on the padding rows of every arrow buffer's interior part and of the border system
__global__ __launch_bounds__(256) void k_arrow_init(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (a < P.ar_nblk) {
    const int row = 6 * P.ar_nint[a] + q;
    if (row < P.ar_nIpad) P.ar_M[(size_t)a * P.ar_ntot * P.ar_ntot + (size_t)row * P.ar_ntot + row] = 1.0;
  } else {
    const int row = 6 * P.ar_nbk + q;
    if (row < P.ar_nb) P.ar_Sb[(size_t)row * P.ar_nb + row] = 1.0;
  }
}

// right-hand side: interior rows into the block's vector, border rows into the border vector (everything else was cleared)
__global__ __launch_bounds__(256) void k_arrow_rhs(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 6 * P.K) return;
  const int pos = q / 6, r = q - 6 * pos, b = P.ar_blk[pos], l = P.ar_loc[pos];
  if (b >= 0) P.ar_rhs[(size_t)b * 2 * P.ar_ntot + 6 * l + r] = P.bp[q];
  else if (b == -1) P.ar_rhsb[6 * l + r] = P.bp[q];
}

// border system += the blocks' Schur contributions (trailing parts of the partially factorised arrow buffers), in block
// order; border right-hand side likewise. One thread per lower-triangle entry.
__global__ __launch_bounds__(256) void k_arrow_border(DevProblem P) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int nbr = 6 * P.ar_nbk;
  if (i >= nbr || j > i) return;
  const int ki = i / 6, ri = i - 6 * ki, kj = j / 6, rj = j - 6 * kj;
  const size_t nt = (size_t)P.ar_ntot;
  double v = P.ar_Sb[(size_t)i * P.ar_nb + j], rv = 0.0;
  for (int a = 0; a < P.ar_nblk; ++a) {
    const int oi = P.ar_own[(size_t)a * P.ar_nbk + ki], oj = P.ar_own[(size_t)a * P.ar_nbk + kj];
    if (oi < 0 || oj < 0) continue;
    v += P.ar_M[(size_t)a * nt * nt + (size_t)(P.ar_nIpad + 6 * oi + ri) * nt + (P.ar_nIpad + 6 * oj + rj)];
  }
  P.ar_Sb[(size_t)i * P.ar_nb + j] = v;
  if (j == i) {  // one thread per row also folds the right-hand side
    rv = P.ar_rhsb[i];
    for (int a = 0; a < P.ar_nblk; ++a) {
      const int oi = P.ar_own[(size_t)a * P.ar_nbk + ki];
      if (oi >= 0) rv += P.ar_rhs[(size_t)a * 2 * nt + P.ar_nIpad + 6 * oi + ri];
    }
    P.ar_rhsb[i] = rv;
  }
}

// border solution -> solution vector and the given-x part of every block that has the keyframe in its own border
__global__ __launch_bounds__(256) void k_arrow_put_border(DevProblem P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (i >= 6 * P.ar_nbk) return;
  const int k = i / 6, r = i - 6 * k;
  const double x = P.ar_rhsb[i];
  if (a < P.ar_nblk) {
    const int o = P.ar_own[(size_t)a * P.ar_nbk + k];
    if (o >= 0) P.ar_rhs[(size_t)a * 2 * P.ar_ntot + P.ar_nIpad + 6 * o + r] = x;
  } else {
    P.bp[6 * P.ar_bpos[k] + r] = x;
  }
}
__global__ __launch_bounds__(256) void k_arrow_scatter(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 6 * P.K) return;
  const int pos = q / 6, r = q - 6 * pos, b = P.ar_blk[pos];
  if (b >= 0) P.bp[q] = P.ar_rhs[(size_t)b * 2 * P.ar_ntot + 6 * P.ar_loc[pos] + r];
}

// [grad | hdiag] of the border keyframes' pose rows <-> contiguous buffer (all-reduced before the damping is applied)
__global__ __launch_bounds__(256) void k_border_vec(DevProblem P, double* __restrict__ buf, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, nbr = 6 * P.ar_nbk;
  if (i >= nbr) return;
  const int k = i / 6, r = i - 6 * k;
  const size_t q = (size_t)P.D * P.pos_kf[P.ar_bpos[k]] + r;
  if (dir == 0) { buf[i] = P.grad[q]; buf[nbr + i] = P.hdiag[q]; }
  else { P.grad[q] = buf[i]; P.hdiag[q] = buf[nbr + i]; }
}
void launch_border_vec(const DevProblem& P, double* buf, int dir, hipStream_t st) {
  if (P.ar_nbk > 0) hipLaunchKernelGGL(k_border_vec, dim3((6 * P.ar_nbk + 255) / 256), dim3(256), 0, st, P, buf, dir);
}

// sharded solve: the max-reduced pair (gradient max-norm, Cholesky failure flag) travels apart from the summed scalars
__global__ void k_shard_scal(DevProblem P, double* mx, int dir) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (dir == 0) { mx[0] = P.scal[SC_GMAX]; mx[1] = (double)P.flag[0]; P.scal[SC_GMAX] = 0.0; }
  else { P.scal[SC_GMAX] = mx[0]; P.flag[0] = (int)mx[1]; }
}
void launch_shard_scal(const DevProblem& P, double* mx, int dir, hipStream_t st) { hipLaunchKernelGGL(k_shard_scal, dim3(1), dim3(64), 0, st, P, mx, dir); }

void launch_arrow_zero(const DevProblem& P, hipStream_t st) {
  if (P.ar_nblk > 0) hipMemsetAsync(P.ar_M, 0, (size_t)P.ar_nblk * P.ar_ntot * P.ar_ntot * sizeof(double), st);
  hipMemsetAsync(P.ar_Sb, 0, (size_t)P.ar_nb * P.ar_nb * sizeof(double), st);
  const int cnt = std::max(P.ar_nIpad, P.ar_nb);
  hipLaunchKernelGGL(k_arrow_init, dim3((cnt + 255) / 256, P.ar_nblk + 1), dim3(256), 0, st, P);
}

void launch_arrow_solve(const DevProblem& P, hipStream_t st, CholAux& ax) {
  const int nblk = P.ar_nblk, ntot = P.ar_ntot, nIpad = P.ar_nIpad, nb = P.ar_nb, nbr = 6 * P.ar_nbk;
  const DenseBatch bt{nblk, (size_t)ntot * ntot, (size_t)nIpad * kTile, (size_t)2 * ntot, P.ar_live, nIpad / kTile, ax.live_h.empty() ? nullptr : ax.live_h.data()};
  if (nblk > 0) hipMemsetAsync(P.ar_rhs, 0, (size_t)nblk * 2 * ntot * sizeof(double), st);
  hipMemsetAsync(P.ar_rhsb, 0, (size_t)2 * nb * sizeof(double), st);
  hipLaunchKernelGGL(k_arrow_rhs, dim3((6 * P.K + 255) / 256), dim3(256), 0, st, P);
  // eliminate every block's interior (tile columns [0, nIpad/128)); forward substitution rides along
  // (a rank of a sharded solve may own no block at all: it still takes part in the border reduction and solve)
  if (nblk > 0) dense_cholesky_solve_raw(P.ar_M, P.ar_rhs, P.ar_Linv, P.flag, ntot, st, ax, nIpad / kTile, false, bt);
  if (nbr > 0) {
    hipLaunchKernelGGL(k_arrow_border, dim3((nbr + 15) / 16, (nbr + 15) / 16), dim3(256), 0, st, P);
    // multi-GPU: every rank holds the contributions of ITS agents, landmarks and factors to the shared-pose system —
    // one all-reduce of [C_b | b_b] (contiguous) per linear solve, then the border is solved redundantly on every rank
    if (ax.reduce != nullptr) ax.reduce(ax.reduce_ctx, P.ar_Sb, (size_t)nb * nb + nb, 0);
    dense_cholesky_solve_raw(P.ar_Sb, P.ar_rhsb, P.ar_Linvb, P.flag, nb, st, ax);  // x_b in ar_rhsb[0 .. nb)
    hipLaunchKernelGGL(k_arrow_put_border, dim3((nbr + 255) / 256, nblk + 1), dim3(256), 0, st, P);
  }
  if (nblk > 0) dense_backward_solve(P.ar_M, P.ar_rhs, P.ar_Linv, ntot, st, nIpad / kTile, ntot / kTile, bt);
  hipLaunchKernelGGL(k_arrow_scatter, dim3((6 * P.K + 255) / 256), dim3(256), 0, st, P);
}

}  // namespace covgpu
