// k_front.hip — multifrontal FP64 Cholesky of the whole reduced camera system: the linear solve of
// ceres::Solve(SPARSE_SCHUR) (optimization_be.cpp:560-567) in the elimination order of nd_plan.hpp.
//
// The reduced camera system (poses 6 + speed-bias 9 per keyframe, A.6) is sparse: covisibility reaches +-20 keyframes along
// an agent's trajectory, loop closures and fused landmarks add local patches, IMU factors couple chain neighbours. Every node
// of the nested-dissection tree owns a dense FRONT [own unknowns | coupled ancestor unknowns]; the linearisation kernels write
// the system straight into the fronts (nd_entry, common.hpp — no global matrix exists). Per tree level, bottom-up:
//   extend-add   a parent's front += the Schur complements its children left in their trailing blocks (one gather launch:
//                every entry sums its children in child order — deterministic, no atomics)
//   partial factorisation of ALL fronts of the level in one batch on the FP64 matrix cores (k_chol.hip / k_panel.hip; fronts
//                of unequal order share the launches through a per-front (offset, leading dimension) table; the forward
//                substitution rides along)
// then top-down one backward substitution per level with the ancestors' unknowns given. Same arithmetic as a dense Cholesky
// of the 15K-order system up to the elimination order. Replaces round 2's IMU-chain elimination + per-agent dense blocks:
// 5-agent map 1.2e11 -> 2.7e10 flops, single-agent maps no longer factor a dense 6K system.
#include <algorithm>

#include "common.hpp"
#include "nd_plan.hpp"

namespace covgpu {

// ------------------------------------------------------------------------------------------------ host tables
void nd_tables(const NdHostPlan& hp, const int* pos_kf, int D, NdDev& dev) {
  dev = NdDev();
  const int nn = hp.nnodes, K = hp.K;
  // nodes renumbered in level order: a level's slice of nd_ntab is its batch table
  std::vector<int> order, newid(nn);
  for (int l = 0; l < hp.nlev; ++l) for (int n : hp.lev_nodes[l]) order.push_back(n);
  for (int i = 0; i < nn; ++i) newid[order[i]] = i;
  dev.h_vnode.assign(2 * (size_t)K, 0); dev.h_voff.assign(2 * (size_t)K, 0); dev.h_vord.assign(2 * (size_t)K, 0);
  for (int v = 0; v < 2 * K; ++v)
    if (hp.vnode[v] >= 0) { dev.h_vnode[v] = newid[hp.vnode[v]]; dev.h_voff[v] = hp.voff[v]; dev.h_vord[v] = hp.vord[v]; }
  dev.h_lev_node.resize(nn);
  dev.h_ndepth.resize(nn); dev.h_nI.resize(nn); dev.h_ntab.resize(2 * (size_t)nn);
  dev.h_own_dims.resize(nn); dev.h_st_dims.resize(nn); dev.h_own_g.resize(nn); dev.h_st_g.resize(nn);
  dev.lev.resize(hp.nlev);
  size_t moff = 0, roff = 0, loff = 0;
  auto gidx_of = [&](int v, int r) { return D * pos_kf[v >> 1] + ((v & 1) ? 6 + r : r); };
  std::vector<int> ld(nn);
  for (int l = 0, i = 0; l < hp.nlev; ++l) {
    NdLevel& L = dev.lev[l];
    L.n = (int)hp.lev_nodes[l].size(); L.nI = hp.lev_nI[l]; L.first = i; L.ntot = hp.lev_nI[l];
    for (int n : hp.lev_nodes[l]) {
      const int nO = ((hp.st_dims[n] + kTile - 1) / kTile) * kTile;
      ld[i] = L.nI + nO;
      L.ntot = std::max(L.ntot, ld[i]);
      dev.h_lev_node[i] = i;
      dev.h_ndepth[i] = hp.depth[n]; dev.h_nI[i] = L.nI;
      dev.h_ntab[2 * (size_t)i] = (long long)moff; dev.h_ntab[2 * (size_t)i + 1] = ld[i];
      moff += (size_t)ld[i] * ld[i];
      dev.h_own_dims[i] = hp.own_dims[n]; dev.h_st_dims[i] = hp.st_dims[n];
      dev.h_own_g[i] = (int)dev.h_gidx.size();
      for (int v : hp.own[n]) for (int r = 0; r < NdHostPlan::vdim(v); ++r) dev.h_gidx.push_back(gidx_of(v, r));
      dev.h_st_g[i] = (int)dev.h_gidx.size();
      for (int v : hp.strct[n]) for (int r = 0; r < NdHostPlan::vdim(v); ++r) dev.h_gidx.push_back(gidx_of(v, r));
      L.live_h.push_back((hp.own_dims[n] + kTile - 1) / kTile);
      L.live_h.push_back(nO / kTile);
      ++i;
    }
    L.rhs_off = roff;
    for (int k = 0; k < L.n; ++k) dev.h_rhs_node.push_back((int)(roff + (size_t)k * 2 * L.ntot));
    roff += (size_t)L.n * 2 * L.ntot;
    L.linv_off = loff; loff += (size_t)L.n * L.nI * kTile;
  }
  dev.M_elems = moff; dev.rhs_elems = roff; dev.linv_elems = loff;
  // front row of every ancestor variable a node's subtree couples to: nd_fidx[nd_abase[node][depth of the ancestor] + ordinal]
  dev.h_abase.assign((size_t)nn * hp.maxdepth, -1);
  for (int i = 0; i < nn; ++i) {
    const int n = order[i];
    for (int a = hp.parent[n]; a >= 0; a = hp.parent[a]) {
      dev.h_abase[(size_t)i * hp.maxdepth + hp.depth[a]] = (int)dev.h_fidx.size();
      dev.h_fidx.resize(dev.h_fidx.size() + hp.own[a].size(), -1);
    }
    int row = dev.h_nI[i];
    for (int v : hp.strct[n]) {
      const int a = hp.vnode[v];
      dev.h_fidx[dev.h_abase[(size_t)i * hp.maxdepth + hp.depth[a]] + hp.vord[v]] = row;
      row += NdHostPlan::vdim(v);
    }
  }
  // children + the map (parent front row -> child front row) the extend-add gathers through
  dev.h_cptr.assign(nn + 1, 0); dev.h_inv_off.assign(nn, -1);
  for (int i = 0; i < nn; ++i) {
    const int n = order[i];
    dev.h_cptr[i + 1] = dev.h_cptr[i] + (int)hp.child[n].size();
    for (int c : hp.child[n]) dev.h_cidx.push_back(newid[c]);
    const int p = hp.parent[n];
    if (p < 0) continue;
    const int ip = newid[p];
    dev.h_inv_off[i] = (int)dev.h_inv.size();
    dev.h_inv.resize(dev.h_inv.size() + ld[ip], -1);
    int* inv = dev.h_inv.data() + dev.h_inv_off[i];
    int row = dev.h_nI[i];
    for (int v : hp.strct[n]) {
      // the variable's row in the parent's front: own column there, or one of the parent's border rows
      const int prow = hp.vnode[v] == p ? hp.voff[v] : dev.h_fidx[dev.h_abase[(size_t)ip * hp.maxdepth + hp.depth[hp.vnode[v]]] + hp.vord[v]];
      for (int r = 0; r < NdHostPlan::vdim(v); ++r) inv[prow + r] = row + r;
      row += NdHostPlan::vdim(v);
    }
  }
  // extend-add work lists: per level the 64x64 tiles of the parents' fronts that receive something from a child
  for (int l = 0; l < hp.nlev; ++l) {
    NdLevel& L = dev.lev[l];
    L.ext_first = (int)dev.h_ext.size() / 3;
    for (int k = 0; k < L.n; ++k) {
      const int i = L.first + k, n = order[i];
      const int T = (ld[i] + 63) / 64;
      std::vector<char> mark((size_t)T * T, 0);
      for (int c : hp.child[n]) {
        const int ic = newid[c];
        const int* inv = dev.h_inv.data() + dev.h_inv_off[ic];
        std::vector<int> rows;
        for (int t = 0; t < T; ++t) {
          bool any = false;
          for (int r = 64 * t; r < std::min(ld[i], 64 * t + 64) && !any; ++r) any = inv[r] >= 0;
          if (any) rows.push_back(t);
        }
        for (int a : rows) for (int b : rows) if (b <= a) mark[(size_t)a * T + b] = 1;
      }
      for (int a = 0; a < T; ++a) for (int b = 0; b <= a; ++b) if (mark[(size_t)a * T + b]) { dev.h_ext.push_back(i); dev.h_ext.push_back(a); dev.h_ext.push_back(b); }
    }
    L.ext_count = (int)dev.h_ext.size() / 3 - L.ext_first;
    L.own_max = 0;
    for (int n : hp.lev_nodes[l]) L.own_max = std::max(L.own_max, hp.own_dims[n]);
  }
  dev.active = true;
}

// ------------------------------------------------------------------------------------------------ device kernels
struct NdLevArgs {
  int first, n, nI, ntot;
  double* rhs;  // the level's right-hand sides: [n][2 ntot]
  const int *own_dims, *st_dims, *own_g, *st_g, *gidx, *cptr, *cidx, *inv_off, *inv;
};

// clear the tiles the factorisation will touch and put the identity on the interior padding rows. One workgroup per 128x128
// tile; dead tiles exit at once. Touched: the lower triangle over [interior tiles up to the last big panel that holds a real
// column | real border tiles] (a half-padded 256-column panel is read whole by the panel kernels), and the 2x2 diagonal
// tiles of the all-padding panels (k_potrf_panel factors every panel of every front of the batch).
__global__ __launch_bounds__(256) void k_nd_zero(DevProblem P, NdLevArgs a) {
  const int node = a.first + blockIdx.z, tr = blockIdx.y, tc = blockIdx.x;
  if (tc > tr) return;
  const int nIt = a.nI / kTile, lo2 = 2 * ((a.own_dims[node] + 2 * kTile - 1) / (2 * kTile)), lb = (a.st_dims[node] + kTile - 1) / kTile;
  auto live = [&](int t) { return t < lo2 || (t >= nIt && t - nIt < lb); };
  const bool pad_diag = tr >= lo2 && tr < nIt && tc >= (tr & ~1);
  if (!((live(tr) && live(tc)) || pad_diag)) return;
  const size_t ld = (size_t)P.nd_ntab[2 * node + 1];
  double* M = P.nd_M + P.nd_ntab[2 * node] + (size_t)tr * kTile * ld + (size_t)tc * kTile;
  const int own = a.own_dims[node];
  for (int e = threadIdx.x; e < kTile * kTile / 2; e += 256) {
    const int r = e / (kTile / 2), c2 = 2 * (e - r * (kTile / 2));
    double2 v = {0.0, 0.0};
    if (tr == tc && tr < nIt) {  // interior diagonal tile: identity on the padding rows
      const int gr = tr * kTile + r;
      if (gr >= own) { if (c2 == r) v.x = 1.0; if (c2 + 1 == r) v.y = 1.0; }
    }
    *reinterpret_cast<double2*>(M + (size_t)r * ld + c2) = v;
  }
}

// speed-bias part of the system (block-tridiagonal rows Ad / Ae and their couplings Bp / Bs / Bn to the poses, filled by the
// inertial kernels and damped by k_finalize_diag) and the right-hand side -> fronts. One thread per entry.
__global__ __launch_bounds__(256) void k_nd_assemble(DevProblem P, const int* __restrict__ rhs_off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (P.vi) {
    const int pos = t / 324, e = t - 324 * pos;
    if (pos < P.K) {
      const int vs = 2 * pos + 1;
      const bool head = pos == P.pos_chain_begin[pos], tail = pos + 1 == P.pos_chain_end[pos];
      if (e < 81) {
        const int r = e / 9, c = e - 9 * r;
        if (c <= r) *nd_entry(P, vs, vs, r, c) = P.Ad[(size_t)81 * pos + e];
      } else if (e < 162) {
        const int q = e - 81, r = q / 9, c = q - 9 * r;
        if (!head) *nd_entry(P, vs, vs - 2, r, c) = P.Ae[(size_t)81 * pos + q];
      } else {
        const int q = e - 162, w = q / 54, k = q - 54 * w, r = k / 6, c = k - 6 * r;
        if (w == 0) *nd_entry(P, vs, 2 * pos, r, c) = P.Bs[(size_t)54 * pos + k];
        else if (w == 1) { if (!head) *nd_entry(P, vs, 2 * (pos - 1), r, c) = P.Bp[(size_t)54 * pos + k]; }
        else if (!tail) *nd_entry(P, vs, 2 * (pos + 1), r, c) = P.Bn[(size_t)54 * pos + k];
      }
    }
  }
  if (t < P.n) {  // right-hand side: own rows of every front (the border rows start at zero and collect the children's parts)
    const int kf = t / P.D, r = t - kf * P.D, pos = P.perm[kf];
    const int v = r < 6 ? 2 * pos : 2 * pos + 1, rr = r < 6 ? r : r - 6;
    const int node = P.nd_vnode[v];
    P.nd_rhs[rhs_off[node] + P.nd_voff[v] + rr] = P.bred[t];
  }
}

// parent front += children's Schur complements, parent right-hand side += children's reduced right-hand sides, in child order.
// One workgroup per 64x64 tile of a host-built list (only tiles some child contributes to); a thread owns a 4x4 sub-grid.
__global__ __launch_bounds__(256) void k_nd_extend(DevProblem P, NdLevArgs a, const int* __restrict__ rhs_off, const int* __restrict__ work) {
  const int node = work[3 * blockIdx.x], tr = work[3 * blockIdx.x + 1], tc = work[3 * blockIdx.x + 2];
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const size_t ld = (size_t)P.nd_ntab[2 * node + 1];
  double* F = P.nd_M + P.nd_ntab[2 * node];
  const int nrow = (int)ld;
  int rr[4], cc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rr[i] = 64 * tr + ty + 16 * i; cc[i] = 64 * tc + tx + 16 * i; }
  // what the front holds already: only columns of the front's OWN unknowns carry original entries (a border x border entry
  // belongs to an ancestor's front) — those are read up front, beside the index loads, the rest starts from zero
  double v[4][4], rv[4] = {0.0, 0.0, 0.0, 0.0};
  bool hit[4][4];
  const bool own_cols = 64 * tc < a.nI;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[i][j] = (own_cols && rr[i] < nrow && cc[j] <= rr[i]) ? F[(size_t)rr[i] * ld + cc[j]] : 0.0;
      hit[i][j] = false;
    }
  for (int k = a.cptr[node]; k < a.cptr[node + 1]; ++k) {
    const int ch = a.cidx[k];
    const int* inv = a.inv + a.inv_off[ch];
    const size_t ldc = (size_t)P.nd_ntab[2 * ch + 1];
    const double* C = P.nd_M + P.nd_ntab[2 * ch];
    int ir[4], ic[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ir[i] = rr[i] < nrow ? inv[rr[i]] : -1; ic[i] = cc[i] < nrow ? inv[cc[i]] : -1; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ir[i] < 0) continue;
      if (tr == tc && tx == 0) rv[i] += P.nd_rhs[rhs_off[ch] + ir[i]];  // (diagonal tiles: lane tx == 0 of a row folds the right-hand side)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ic[j] < 0 || cc[j] > rr[i]) continue;
        v[i][j] += C[(size_t)(ir[i] > ic[j] ? ir[i] : ic[j]) * ldc + (ir[i] > ic[j] ? ic[j] : ir[i])];
        hit[i][j] = true;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (hit[i][j]) F[(size_t)rr[i] * ld + cc[j]] = v[i][j];
    if (rv[i] != 0.0) P.nd_rhs[rhs_off[node] + rr[i]] += rv[i];
  }
}

// block inverses of the padding columns are the identity once and for all (the panel kernel never writes them: DenseBatch::own_max)
__global__ __launch_bounds__(256) void k_nd_linv_init(double* __restrict__ Linv, size_t n) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) { const int e = (int)(q & 255); Linv[q] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0; }
}

// ------------------------------------------------------------------------------------------------ launchers
static NdLevArgs lev_args(const DevProblem& P, const NdDev& nd, int l) {
  const NdLevel& L = nd.lev[l];
  return NdLevArgs{L.first, L.n, L.nI, L.ntot, P.nd_rhs + L.rhs_off, nd.own_dims, nd.st_dims, nd.own_g, nd.st_g, nd.gidx, nd.cptr, nd.cidx, nd.inv_off, nd.inv};
}

void launch_nd_init(const DevProblem& P, const NdDev& nd, hipStream_t st) {
  if (nd.linv_elems) hipLaunchKernelGGL(k_nd_linv_init, dim3((unsigned)((nd.linv_elems + 255) / 256)), dim3(256), 0, st, P.nd_Linv, nd.linv_elems);
}

void launch_nd_zero(const DevProblem& P, const NdDev& nd, hipStream_t st) {
  for (size_t l = 0; l < nd.lev.size(); ++l) {
    const NdLevel& L = nd.lev[l];
    const int T = L.ntot / kTile;
    hipLaunchKernelGGL(k_nd_zero, dim3(T, T, L.n), dim3(256), 0, st, P, lev_args(P, nd, (int)l));
  }
}

void launch_nd_solve(const DevProblem& P, NdDev& nd, double* dst, hipStream_t st, CholAux& ax) {
  const int nlev = (int)nd.lev.size();
  ax.mark(st, -1);
  hipMemsetAsync(P.nd_rhs, 0, nd.rhs_elems * sizeof(double), st);
  {
    const int cnt = std::max(P.vi ? 324 * P.K : 0, P.n);
    hipLaunchKernelGGL(k_nd_assemble, dim3((cnt + 255) / 256), dim3(256), 0, st, P, (const int*)nd.rhs_node);
  }
  ax.mark(st, -2);
  auto batch = [&](int l) {
    const NdLevel& L = nd.lev[l];
    DenseBatch bt;
    bt.n = L.n; bt.sM = 0; bt.sL = (size_t)L.nI * kTile; bt.sR = (size_t)2 * L.ntot;
    bt.live = L.live; bt.tI = L.nI / kTile; bt.live_h = L.live_h.data();
    bt.tab = P.nd_ntab + 2 * (size_t)L.first; bt.tri_slot = l; bt.own_max = L.own_max;
    return bt;
  };
  for (int l = 0; l < nlev; ++l) {
    const NdLevel& L = nd.lev[l];
    if (L.ext_count > 0)
      hipLaunchKernelGGL(k_nd_extend, dim3(L.ext_count), dim3(256), 0, st, P, lev_args(P, nd, l), (const int*)nd.rhs_node, (const int*)(nd.ext + 3 * (size_t)L.ext_first));
    ax.mark(st, -3);
    dense_cholesky_solve_raw(P.nd_M, P.nd_rhs + L.rhs_off, P.nd_Linv + L.linv_off, P.flag, L.ntot, st, ax, L.nI / kTile, false, batch(l));
    ax.mark(st, -4);
  }
  for (int l = nlev - 1; l >= 0; --l) {
    // top-down: the ancestors' unknowns are read from `dst` by the first launch of the level, the fronts' own unknowns are
    // written there by its last (BwdXfer)
    const NdLevel& L = nd.lev[l];
    DenseBatch bt = batch(l);
    bt.xfer.gidx = nd.gidx; bt.xfer.own_g = nd.own_g; bt.xfer.st_g = nd.st_g; bt.xfer.own_dims = nd.own_dims; bt.xfer.st_dims = nd.st_dims;
    bt.xfer.x = dst; bt.xfer.first = L.first;
    dense_backward_solve(P.nd_M, P.nd_rhs + L.rhs_off, P.nd_Linv + L.linv_off, L.ntot, st, L.nI / kTile, L.ntot / kTile, bt);
    ax.mark(st, -5);
  }
  ax.mark(st, -6);
}

}  // namespace covgpu
