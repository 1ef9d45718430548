// k_pgo.hip — block-arrow solve of the pose-graph system (PoseGraphOptimization, optimization_be.cpp:1024-1031).
//
// The pose-graph Hessian is not dense: every keyframe is tied to at most its five predecessors (optimization_be.cpp:
// 947-1021), agents are coupled with each other only through loop edges, and there are few of those (:912-944).
// Factorising it densely costs what a GBA iteration costs — bound by the serial panel chain of the 6K-order matrix
// (103 potrf steps, 25 ms on the 5-agent map) although almost every tile is zero. Instead:
//   border   = keyframes at the ends of loop edges                                             -> last
//   blocks   = connected components of the rest (the agents), merged to at most kPgoMaxBlocks  -> independent
// H = [ D_0        B_0 ]   each block's arrow [D_a B_a; B_a^T 0] is copied into its own dense buffer and its first
//     [     D_1    B_1 ]   T_a tile columns are eliminated by the SAME MFMA Cholesky (partial factorisation, k_chol.hip),
//     [ B_0^T B_1^T  C ]   all blocks batched into the same launches: the trailing block of each buffer then holds
// -X_b X_b^T, the block's Schur contribution, and the trailing part of its right-hand side -X_b y_a. The border system
// C - sum_a X_b X_b^T (order 6 x #border keyframes) is summed in a fixed order, solved densely, and every block
// finishes with its own backward substitution. Same arithmetic as the dense solve up to the elimination order.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"

namespace covgpu {

constexpr int kPgoMaxBlocks = 32;   // blocks of one batched factorisation

// host: pick the border and cut the rest into independent blocks, from the edge list alone (the map hands keyframes
// over agent-interleaved, typedefs_base.hpp:178, so index distance says nothing). An edge whose endpoints have no
// common neighbour is a bridge between otherwise unrelated parts of the graph — a loop closure; the odometry edges of
// one agent (each keyframe tied to its five predecessors) always share neighbours. Border = endpoints of such
// edges; blocks = connected components of what is left (union-find), merged to at most kPgoMaxBlocks. Any partition
// found this way is valid (no interior-interior edge crosses blocks by construction); the heuristic only decides
// whether it pays. Returns false if it does not (no split, or a large border): the caller keeps the dense solve.
bool pgo_plan_analyse(int K, int E, const int* ei, const int* ej, PgoHostPlan& out) {
  out = PgoHostPlan();
  if (6 * K < 8 * kTile) return false;  // fewer than eight tiles of unknowns: the dense solve is a handful of launches
  std::vector<std::vector<int>> adj(K);
  for (int e = 0; e < E; ++e) { adj[ei[e]].push_back(ej[e]); adj[ej[e]].push_back(ei[e]); }
  for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  std::vector<char> bridge_border(K, 0);
  for (int e = 0; e < E; ++e) {
    const std::vector<int>&A = adj[ei[e]], &B = adj[ej[e]];
    bool common = false;
    for (size_t x = 0, y = 0; x < A.size() && y < B.size();) {
      if (A[x] == B[y]) { common = true; break; }
      if (A[x] < B[y]) ++x; else ++y;
    }
    if (!common && (A.size() > 1 || B.size() > 1)) { bridge_border[ei[e]] = 1; bridge_border[ej[e]] = 1; }
  }
  // One candidate plan per segment length: long components (an agent's whole trajectory) are cut further — in a
  // breadth-first level structure edges join equal or adjacent levels only, so removing one whole level separates
  // what came before from what comes after. Levels of an odometry chain are ~5 keyframes wide; one level per ~segment
  // keyframes goes to the border. The serial panel chain of the batched factorisation is as long as the LARGEST
  // block, the border system grows with every cut, and all buffers are re-gathered every iteration; the cheapest
  // candidate under that model (fitted on the 5-agent map: 0.35 ms per 256-wide panel step, 2.5 ms per GB of arrow
  // buffers; segment lengths 40..250 measured 84, 95, 65, 73, 83 ms per call) is taken.
  std::vector<int> parent(K), level(K, -1), queue;
  auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  auto build = [&](int segment, PgoHostPlan& plan, double& cost) -> bool {
    std::vector<char> border = bridge_border;
    std::vector<std::vector<int>> comps;
    int n_int = 0;
    auto components = [&]() {
      for (int k = 0; k < K; ++k) parent[k] = k;
      for (int e = 0; e < E; ++e)
        if (!border[ei[e]] && !border[ej[e]]) { const int a = find(ei[e]), b = find(ej[e]); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
      comps.clear();
      std::vector<int> comp_of(K, -1);
      n_int = 0;
      for (int k = 0; k < K; ++k) {
        if (border[k]) continue;
        const int r = find(k);
        if (comp_of[r] < 0) { comp_of[r] = (int)comps.size(); comps.emplace_back(); }
        comps[comp_of[r]].push_back(k);
        ++n_int;
      }
    };
    components();
    auto bfs = [&](int start) {  // levels inside start's component (border excluded); returns the last node reached
      queue.assign(1, start);
      level[start] = 0;
      for (size_t h = 0; h < queue.size(); ++h) {
        const int u = queue[h];
        for (int v : adj[u]) if (!border[v] && level[v] < 0) { level[v] = level[u] + 1; queue.push_back(v); }
      }
      return queue.back();
    };
    const std::vector<std::vector<int>> first = comps;
    for (const auto& cmp : first) {
      if ((int)cmp.size() < 2 * segment) continue;
      const int far = bfs(cmp[0]);
      for (int v : queue) level[v] = -1;
      bfs(far);  // second sweep from a pseudo-peripheral node
      const std::vector<int> order = queue;
      int since = 0, cut_level = -1, prev_level = -1, remaining = (int)order.size();
      for (int v : order) {  // breadth-first order: levels arrive one after the other
        if (level[v] != prev_level) {  // first keyframe of a new level: the only place a cut may start
          prev_level = level[v];
          if (since >= segment && remaining > segment / 2) { cut_level = level[v]; since = 0; }
        }
        --remaining;
        if (level[v] == cut_level) border[v] = 1;
        else ++since;
      }
      for (int v : order) level[v] = -1;
    }
    components();
    const int n_border = K - n_int;
    if (comps.size() < 2 || n_border * 3 > K) return false;
    // largest components first; the rest is packed onto the currently smallest block
    std::sort(comps.begin(), comps.end(), [](const std::vector<int>& x, const std::vector<int>& y) { return x.size() != y.size() ? x.size() > y.size() : x[0] < y[0]; });
    if ((int)comps[0].size() * 10 > n_int * 7) return false;  // one component dominates: its chain of panels is the whole cost anyway
    std::vector<std::vector<int>> blocks;
    for (auto& cmp : comps) {
      if ((int)blocks.size() < kPgoMaxBlocks) { blocks.push_back(cmp); continue; }
      size_t best = 0;
      for (size_t q = 1; q < blocks.size(); ++q) if (blocks[q].size() < blocks[best].size()) best = q;
      blocks[best].insert(blocks[best].end(), cmp.begin(), cmp.end());
    }
    size_t big = 0;
    for (auto& bl : blocks) { std::sort(bl.begin(), bl.end()); big = std::max(big, bl.size()); }
    plan = PgoHostPlan();
    for (int k = 0; k < K; ++k) if (border[k]) plan.border_kf.push_back(k);
    const double nI = std::ceil(6.0 * big / (2 * kTile)) * (2 * kTile), nb = std::max(1.0, std::ceil(6.0 * n_border / kTile)) * kTile;
    cost = 0.35 * (nI + nb) / (2 * kTile) + 2.5 * (double)blocks.size() * (nI + nb) * (nI + nb) * 8e-9;
    plan.block_kf = std::move(blocks);
    return true;
  };
  bool have = false;
  double best_cost = 0.0;
  for (int segment : {60, 80, 100, 130, 170, 220, 300, 1 << 30}) {  // the last one: whole components, no cuts
    PgoHostPlan cand;
    double cost = 0.0;
    if (!build(segment, cand, cost)) continue;
    if (!have || cost < best_cost) { out = std::move(cand); best_cost = cost; have = true; }
  }
  return have;
}

// arrow buffers, one per block (blockIdx.z), all of the same padded shape: M[i][j] = H[idx[i]][idx[j]] (H symmetric,
// lower stored), identity on padding, ZERO in the trailing (border x border) block; rhs[i] = b[idx[i]] on the block's
// own rows, zero on the border rows
__global__ __launch_bounds__(256) void k_pgo_gather(const double* __restrict__ H, size_t ldh, const double* __restrict__ bp,
                                                     const int* __restrict__ idx_all, int ntot, int nIpad, double* __restrict__ M_all,
                                                     double* __restrict__ rhs_all) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4), a = blockIdx.z;
  if (i >= ntot || j >= ntot) return;
  const int* idx = idx_all + (size_t)a * ntot;
  const int gi = idx[i], gj = idx[j];
  double v;
  if (i >= nIpad && j >= nIpad) v = 0.0;
  else if (gi < 0 || gj < 0) v = (i == j) ? 1.0 : 0.0;
  else v = H[(size_t)max(gi, gj) * ldh + min(gi, gj)];
  M_all[(size_t)a * ntot * ntot + (size_t)i * ntot + j] = v;
  if (j == 0) {
    double* rhs = rhs_all + (size_t)a * 2 * ntot;
    rhs[i] = (i < nIpad && gi >= 0) ? bp[gi] : 0.0;
    rhs[ntot + i] = 0.0;
  }
}

// border system: C - sum_a X_b X_b^T and b_b - sum_a X_b y_a, blocks added in index order
__global__ __launch_bounds__(256) void k_pgo_border(const double* __restrict__ H, size_t ldh, const double* __restrict__ bp,
                                                     const int* __restrict__ idxb, int nb, const double* __restrict__ M_all,
                                                     const double* __restrict__ rhs_all, int nblk, int ntot, int nIpad,
                                                     double* __restrict__ Sb, double* __restrict__ rhsb) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= nb || j > i) return;
  const int gi = idxb[i], gj = idxb[j];
  double v;
  if (gi < 0 || gj < 0) v = (i == j) ? 1.0 : 0.0;
  else {
    v = H[(size_t)max(gi, gj) * ldh + min(gi, gj)];
    for (int a = 0; a < nblk; ++a) v += M_all[(size_t)a * ntot * ntot + (size_t)(nIpad + i) * ntot + nIpad + j];
  }
  Sb[(size_t)i * nb + j] = v;
  Sb[(size_t)j * nb + i] = v;
  if (j == 0) {
    double r = 0.0;
    if (gi >= 0) {
      r = bp[gi];
      for (int a = 0; a < nblk; ++a) r += rhs_all[(size_t)a * 2 * ntot + nIpad + i];
    }
    rhsb[i] = r;
  }
}

// x_b -> solution vector (blockIdx.y == nblk) and into the given-x part of every block's vector (blockIdx.y = block)
__global__ __launch_bounds__(256) void k_pgo_put_border(const double* __restrict__ xb, const int* __restrict__ idxb, int nb, double* __restrict__ sol,
                                                         double* __restrict__ rhs_all, int nblk, int ntot, int nIpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (i >= nb) return;
  if (a < nblk) rhs_all[(size_t)a * 2 * ntot + nIpad + i] = xb[i];
  else if (idxb[i] >= 0) sol[idxb[i]] = xb[i];
}
__global__ __launch_bounds__(256) void k_pgo_scatter(const double* __restrict__ rhs_all, const int* __restrict__ idx_all, int ntot, int nIpad,
                                                      double* __restrict__ sol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (i >= nIpad) return;
  const int gi = idx_all[(size_t)a * ntot + i];
  if (gi >= 0) sol[gi] = rhs_all[(size_t)a * 2 * ntot + i];
}

// Solves H x = b for the pose-graph system assembled in P.Sred / P.bp (map order); x overwrites P.bp.
// All blocks share one shape (padded to the largest) and advance in lockstep through ONE batched partial factorisation
// on the context's usual four streams: the same ~300 launches as for a single block, each covering every block
// (one more grid dimension). (Separate stream sets per block, each with its own launches, were host-bound and
// multiplexed onto the runtime's four hardware queues: 23.5 ms per iteration against 25.6 dense.)
void launch_pgo_block_solve(const DevProblem& P, PgoPlan& plan, hipStream_t st, CholAux& ax) {
  const size_t ldh = (size_t)P.npad;
  const int nblk = plan.nblk, ntot = plan.ntot, nIpad = plan.nIpad, nb = plan.nb;
  const DenseBatch bt{nblk, (size_t)ntot * ntot, (size_t)nIpad * kTile, (size_t)2 * ntot};
  hipLaunchKernelGGL(k_pgo_gather, dim3((ntot + 15) / 16, (ntot + 15) / 16, nblk), dim3(256), 0, st, P.Sred, ldh, P.bp, plan.idx, ntot, nIpad, plan.M,
                     plan.rhs);
  dense_cholesky_solve_raw(plan.M, plan.rhs, plan.Linv, P.flag, ntot, st, ax, nIpad / kTile, false, bt);
  hipLaunchKernelGGL(k_pgo_border, dim3((nb + 15) / 16, (nb + 15) / 16), dim3(256), 0, st, P.Sred, ldh, P.bp, plan.idx_b, nb, plan.M, plan.rhs, nblk, ntot,
                     nIpad, plan.Sb, plan.rhs_b);
  dense_cholesky_solve_raw(plan.Sb, plan.rhs_b, plan.Linv_b, P.flag, nb, st, ax);  // x_b in rhs_b[0 .. nb)
  hipLaunchKernelGGL(k_pgo_put_border, dim3((nb + 255) / 256, nblk + 1), dim3(256), 0, st, plan.rhs_b, plan.idx_b, nb, P.bp, plan.rhs, nblk, ntot, nIpad);
  dense_backward_solve(plan.M, plan.rhs, plan.Linv, ntot, st, nIpad / kTile, ntot / kTile, bt);
  hipLaunchKernelGGL(k_pgo_scatter, dim3((nIpad + 255) / 256, nblk), dim3(256), 0, st, plan.rhs, plan.idx, ntot, nIpad, P.bp);
}

// right-hand side of the pose-graph system into bp (padded), solution back into the IR layout
__global__ __launch_bounds__(256) void k_pg_gather_rhs(DevProblem P) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.npad) return;
  double v = 0.0;
  if (q < 6 * P.K) { const int pos = q / 6, r = q - 6 * pos; v = P.bred[(size_t)P.D * P.pos_kf[pos] + r]; }
  P.bp[q] = v;
}
__global__ __launch_bounds__(256) void k_pg_scatter_solution(DevProblem P, double* __restrict__ dst) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.n) return;
  const int kf = q / P.D, r = q - kf * P.D;
  dst[q] = P.bp[6 * P.perm[kf] + r];
}

// PoseGraphOptimization's linear solve (ceres::Solve at optimization_be.cpp:1024-1031: no landmarks, so SPARSE_SCHUR
// degenerates to a sparse Cholesky on the poses): block-arrow elimination when a plan exists, else the plain dense Cholesky
void launch_pose_graph_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax, PgoPlan* pgo) {
  hipLaunchKernelGGL(k_pg_gather_rhs, dim3((P.npad + 255) / 256), dim3(256), 0, st, P);
  if (pgo != nullptr) launch_pgo_block_solve(P, *pgo, st, ax);
  else dense_cholesky_solve_raw(P.Sred, P.bp, P.Linv, P.flag, P.npad, st, ax);
  hipLaunchKernelGGL(k_pg_scatter_solution, dim3((P.n + 255) / 256), dim3(256), 0, st, P, dst);
}

}  // namespace covgpu
