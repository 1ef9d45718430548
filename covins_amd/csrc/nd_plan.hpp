// nd_plan.hpp — host-side plan of the multifrontal (nested-dissection) Cholesky of the reduced camera system
// (k_front.hip). Replaces the elimination order ceres::Solve(SPARSE_SCHUR) gets from CHOLMOD's fill-reducing
// ordering (optimization_be.cpp:560-565).
//
// Unknowns ("variables") of the reduced camera system after landmark elimination: per keyframe position q a 6-dim pose
// block P_q (variable 2q) and — visual-inertial only — a 9-dim speed-bias block V_q (variable 2q+1). Structural couplings:
//   P_i ~ P_j   covisible keyframes (common landmark, A.6) and loop edges (optimization_be.cpp:538-556)
//   IMU factor between chain neighbours (q-1, q): all pairs of {P_q-1, V_q-1, P_q, V_q} (optimization_be.cpp:415-416)
// The plan is a tree of supernodes found by recursive bisection (agents first, then along each agent's time axis): a
// vertex cover of the couplings crossing a cut is the cut's separator node, the two halves recurse below it. Every node's
// front is [own variables | the ancestor variables its subtree couples to]; fronts of equal height form one batch.
#pragma once
#include <cstdint>
#include <vector>

namespace covgpu {

struct NdHostPlan {
  int K = 0, vi = 0, nvar = 0;
  int nnodes = 0, nlev = 0, maxdepth = 0;
  std::vector<int> parent, depth, level, slot;      // per node: parent (-1 root) | depth (root 0) | height = batch level | index in its batch
  std::vector<std::vector<int>> own, strct, child;  // per node: own variables (elimination order) | ancestor variables of the front | children
  std::vector<int> own_dims, st_dims;                // per node: scalar sizes
  std::vector<int> vnode, voff, vord;                // per variable: node | scalar offset inside the node's own columns | ordinal inside the node (-1: absent)
  std::vector<std::vector<int>> lev_nodes;           // per level: its nodes, batch order
  std::vector<int> lev_nI, lev_nO, lev_ntot;         // per level: padded interior order (multiple of 256) | padded border order (multiple of 128) | sum
  // agent-sharded solve of one map over several GPUs (nd_shard_assign): per node the owning rank, -1 = TOP node (replicated:
  // every rank holds its front, the ranks' contributions are all-reduced once per linear solve). Empty: single GPU.
  std::vector<int> node_rank;
  int nsub = 0;                                      // subtrees hanging below the top nodes (the units dealt to the ranks)
  int top_mode = 0, leaf = 0, group_frac100 = 300;   // which candidate nd_plan_build kept: cut of a region of >= 3 agents (0 one cover | 1 two groups) | leaf size | 100 x the two groups' balance bound
  double flops = 0;                                  // partial factorisations, dense count on the real (unpadded) sizes
  size_t front_elems = 0;                            // sum over levels of batch x ntot^2
  static int vdim(int v) { return (v & 1) ? 9 : 6; }
};

// positions are chain-major (solver.hip build_chains); pair / epair lists are keyframe pairs by position (i > j).
// leaf_dims: a region of at most this many scalar unknowns is not cut further. Returns false on an inconsistent input.
bool nd_plan_build(int K, bool vi, int nchains, const int* chain_ptr, int npairs, const int* pair_i, const int* pair_j, int nepairs,
                   const int* epair_i, const int* epair_j, int leaf_dims, NdHostPlan& out, int top_mode = -1);
// top_mode: how a region of three or more agents is cut — 0: ONE cover of all cross-agent couplings (its children: one region per agent;
// the plan of a sharded solve) | 1: two groups of agents, recursively | -1: both are built and the cheaper one is kept (nd_plan.hip).

// Multi-GPU split of the tree: the top of the tree is replicated, the subtrees below it are dealt to the
// ranks by longest-processing-time-first on their factorisation flops. Top = the roots, grown downwards (heaviest subtree
// first) until there are at least `world` subtrees and none outweighs 1.25x a rank's fair share — but never beyond 48 MiB of top
// fronts (every top front is all-reduced and factorised on every rank): rather fewer subtrees than ranks. Deterministic.
void nd_shard_assign(NdHostPlan& hp, int world, double top_cap_bytes = 48.0 * 1048576.0);
// what one factorisation costs a rank of a sharded solve, roughly: (replicated top + the busiest rank's subtrees) at 30 TFLOP/s, the serial panel chains of
// every level, and the ring all-reduce of the top fronts' lower halves at 150 GB/s per link. Compares candidate shard plans (covgpu_shard_plan).
double nd_shard_cost(const NdHostPlan& hp, int world);
// variable ids -> var_map[old id] (chain positions of one problem -> IR keyframes -> chain positions of a rank's sub-problem)
void nd_plan_remap(NdHostPlan& hp, const std::vector<int>& var_map);

}  // namespace covgpu
