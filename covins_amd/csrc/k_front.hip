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
#include "dev_math.hpp"
#include "nd_plan.hpp"

namespace covgpu {

// ------------------------------------------------------------------------------------------------ host tables
// rank: this context's rank in an agent-sharded solve (hp.node_rank non-empty); fronts exist for the rank's own subtrees and for
// the TOP nodes (replicated). Batches = levels: subtree nodes by height, then the top nodes by their height inside the top.
void nd_tables(const NdHostPlan& hp, const int* pos_kf, int D, int rank, NdDev& dev) {
  dev = NdDev();
  const int nn = hp.nnodes, K = hp.K;
  const bool sh = !hp.node_rank.empty();
  auto is_top = [&](int n) { return sh && hp.node_rank[n] < 0; };
  auto is_local = [&](int n) { return !sh || hp.node_rank[n] < 0 || hp.node_rank[n] == rank; };
  int Hs = 0;
  for (int n = 0; n < nn; ++n) if (!is_top(n)) Hs = std::max(Hs, hp.level[n] + 1);
  std::vector<int> lev(nn), toph(nn, 0);
  for (int n = nn - 1; n >= 0; --n) if (is_top(n)) for (int c : hp.child[n]) if (is_top(c)) toph[n] = std::max(toph[n], toph[c] + 1);
  int nlev = Hs;
  for (int n = 0; n < nn; ++n) { lev[n] = is_top(n) ? Hs + toph[n] : hp.level[n]; nlev = std::max(nlev, lev[n] + 1); }
  std::vector<std::vector<int>> lnodes(nlev);
  for (int n = 0; n < nn; ++n) if (is_local(n)) lnodes[lev[n]].push_back(n);
  // local nodes renumbered in level order: a level's slice of nd_ntab is its batch table; foreign nodes have no id
  std::vector<int> order, newid(nn, -1);
  for (int l = 0; l < nlev; ++l) for (int n : lnodes[l]) { newid[n] = (int)order.size(); order.push_back(n); }
  const int nl = (int)order.size();
  dev.nnodes = nl; dev.top_lev0 = Hs; dev.plan_flops = hp.flops;
  dev.h_vnode.assign(2 * (size_t)K, -1); dev.h_voff.assign(2 * (size_t)K, 0); dev.h_vord.assign(2 * (size_t)K, 0); dev.h_vown.assign(2 * (size_t)K, 0);
  for (int v = 0; v < 2 * K; ++v)
    if (hp.vnode[v] >= 0) {
      const int n = hp.vnode[v];
      dev.h_vnode[v] = newid[n]; dev.h_voff[v] = hp.voff[v]; dev.h_vord[v] = hp.vord[v];
      dev.h_vown[v] = !is_local(n) ? 0 : (is_top(n) ? 2 : 1);
    }
  dev.h_ndepth.resize(nl); dev.h_nI.resize(nl); dev.h_ntab.resize(2 * (size_t)nl);
  dev.h_own_dims.resize(nl); dev.h_st_dims.resize(nl); dev.h_own_g.resize(nl); dev.h_st_g.resize(nl);
  dev.lev.resize(nlev);
  auto gidx_of = [&](int v, int r) { return D * pos_kf[v >> 1] + ((v & 1) ? 6 + r : r); };
  std::vector<int> ld(nl);
  size_t moff = 0, loff = 0;
  for (int l = 0, i = 0; l < nlev; ++l) {
    NdLevel& L = dev.lev[l];
    L.n = (int)lnodes[l].size(); L.first = i; L.nI = 256; L.own_max = 0;
    for (int n : lnodes[l]) L.own_max = std::max(L.own_max, hp.own_dims[n]);
    L.nI = std::max(256, ((L.own_max + 255) / 256) * 256);
    L.ntot = L.nI;
    if (l == Hs) dev.M_sub = moff;
    for (int n : lnodes[l]) {
      const int nO = ((hp.st_dims[n] + kTile - 1) / kTile) * kTile;
      ld[i] = L.nI + nO;
      L.ntot = std::max(L.ntot, ld[i]);
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
      if (is_top(n)) {   // live lower-triangular tiles of a replicated front: what a sharded solve exchanges (same liveness as k_nd_zero)
        const int nIt = L.nI / kTile, lo2 = 2 * ((hp.own_dims[n] + 2 * kTile - 1) / (2 * kTile)), lb = (hp.st_dims[n] + kTile - 1) / kTile;
        auto live = [&](int t) { return t < lo2 || (t >= nIt && t - nIt < lb); };
        const int Tn = ld[i] / kTile;
        for (int tr = 0; tr < Tn; ++tr) for (int tc = 0; tc <= tr; ++tc) if (live(tr) && live(tc)) { dev.h_top_tiles.push_back(i); dev.h_top_tiles.push_back(tr); dev.h_top_tiles.push_back(tc); }
      }
      if (is_top(n))
        for (int v : hp.own[n]) for (int r = 0; r < NdHostPlan::vdim(v); ++r) { dev.h_top_var.push_back(v); dev.h_top_r.push_back(r); dev.h_top_g.push_back(gidx_of(v, r)); }
      ++i;
    }
    L.linv_off = loff; loff += (size_t)L.n * L.nI * kTile;
    {  // front lists of the panel factorisation (NdLevel::plist)
      L.plist_h.resize(L.n);
      for (int k = 0; k < L.n; ++k) L.plist_h[k] = k;
      std::stable_sort(L.plist_h.begin(), L.plist_h.end(), [&](int a, int b) { return dev.h_own_dims[L.first + a] > dev.h_own_dims[L.first + b]; });
      const int np = L.nI / 256;
      L.pbig.assign(np, 0); L.psmall.assign(np, 0);
      for (int k = 0; k < L.n; ++k)
        for (int P = 0; P < np; ++P) {
          const int real = dev.h_own_dims[L.first + k] - 256 * P;
          if (real > 128) ++L.pbig[P]; else if (real > 0) ++L.psmall[P];
        }
    }
  }
  if (Hs >= nlev) dev.M_sub = moff;
  dev.M_elems = moff; dev.linv_elems = loff;
  // right-hand sides: [top levels | grad, hdiag of the top unknowns | subtree levels] — with the top fronts right in front of them
  // (solver.hip puts both in ONE allocation) the whole all-reduced region is contiguous
  dev.h_rhs_node.assign(nl, 0);
  size_t roff = 0;
  auto place = [&](int l) {
    NdLevel& L = dev.lev[l];
    L.rhs_off = roff;
    for (int k = 0; k < L.n; ++k) dev.h_rhs_node[L.first + k] = (int)(roff + (size_t)k * 2 * L.ntot);
    roff += (size_t)L.n * 2 * L.ntot;
  };
  for (int l = Hs; l < nlev; ++l) place(l);
  dev.rhs_top = roff; dev.gh_off = roff; roff += 2 * dev.h_top_g.size();
  for (int l = 0; l < Hs; ++l) place(l);
  dev.rhs_elems = roff;
  // front row of every ancestor variable a node's subtree couples to: nd_fidx[nd_abase[node][depth of the ancestor] + ordinal]
  dev.h_abase.assign((size_t)nl * hp.maxdepth, -1);
  for (int i = 0; i < nl; ++i) {
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
  // children (subtree children | top children, see NdDev) + the map (parent front row -> child front row) the extend-add gathers through
  dev.h_cptr.assign(nl + 1, 0); dev.h_cptr2.assign(nl + 1, 0); dev.h_inv_off.assign(nl, -1);
  for (int i = 0; i < nl; ++i) {
    const int n = order[i];
    dev.h_cptr[i + 1] = dev.h_cptr[i]; dev.h_cptr2[i + 1] = dev.h_cptr2[i];
    for (int c : hp.child[n]) {
      if (!is_local(c)) continue;   // another rank's subtree: its contribution arrives through the all-reduce of the top fronts
      if (is_top(c)) { dev.h_cidx2.push_back(newid[c]); ++dev.h_cptr2[i + 1]; }
      else { dev.h_cidx.push_back(newid[c]); ++dev.h_cptr[i + 1]; }
    }
    const int p = hp.parent[n];
    if (p < 0) continue;
    const int ip = newid[p];   // (a local node's parent is local: own subtree or top)
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
  // extend-add work lists: per level the 64x64 tiles of the parents' fronts that receive something from a child of either kind
  auto worklist = [&](int l, const std::vector<int>& cptr, const std::vector<int>& cidx, int& first, int& count, int& countA) {
    NdLevel& L = dev.lev[l];
    first = (int)dev.h_ext.size() / 3;
    std::vector<int> tiles;
    for (int k = 0; k < L.n; ++k) {
      const int i = L.first + k;
      const int T = (ld[i] + 63) / 64;
      std::vector<char> mark((size_t)T * T, 0);
      for (int q = cptr[i]; q < cptr[i + 1]; ++q) {
        const int* inv = dev.h_inv.data() + dev.h_inv_off[cidx[q]];
        std::vector<int> rows;
        for (int t = 0; t < T; ++t) {
          bool any = false;
          for (int r = 64 * t; r < std::min(ld[i], 64 * t + 64) && !any; ++r) any = inv[r] >= 0;
          if (any) rows.push_back(t);
        }
        for (int a : rows) for (int b : rows) if (b <= a) mark[(size_t)a * T + b] = 1;
      }
      // Round 6: the border x border part of a front is never cleared (k_nd_zero) — the extend-add STORES every entry of the 64-tiles it visits there
      // (children's sum, or zero), so it must visit every lower 64-tile of each 128-tile some child reaches (NdDev::bb marks those for the first
      // trailing update, which reads them; the others it starts from zero itself)
      if (&cptr == &dev.h_cptr && !is_top(order[i])) {
        const int t0 = dev.h_nI[i] / 64;
        for (int A = t0; A < T; A += 2)
          for (int B = t0; B <= A; B += 2) {
            bool any = false;
            for (int a = A; a < std::min(T, A + 2); ++a) for (int b = B; b < std::min(T, B + 2); ++b) if (b <= a && mark[(size_t)a * T + b]) any = true;
            if (any) for (int a = A; a < std::min(T, A + 2); ++a) for (int b = B; b < std::min(T, B + 2); ++b) if (b <= a) mark[(size_t)a * T + b] = 1;
          }
      }
      for (int a = 0; a < T; ++a) for (int b = 0; b <= a; ++b) if (mark[(size_t)a * T + b]) { tiles.push_back(i); tiles.push_back(a); tiles.push_back(b); }
    }
    // tiles of the fronts' first 256 rows (what the first panel's factorisation needs) first: NdLevel::ext_countA
    countA = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (size_t q = 0; q < tiles.size(); q += 3)
        if ((tiles[q + 1] < 4) == (pass == 0)) {
          dev.h_ext.insert(dev.h_ext.end(), tiles.begin() + q, tiles.begin() + q + 3); countA += pass == 0 ? 1 : 0;
          dev.h_ext_kind.push_back(&cptr == &dev.h_cptr2 ? 1 : 0);
        }
    count = (int)dev.h_ext.size() / 3 - first;
  };
  for (int l = 0; l < nlev; ++l) {
    worklist(l, dev.h_cptr, dev.h_cidx, dev.lev[l].ext_first, dev.lev[l].ext_count, dev.lev[l].ext_countA);
    worklist(l, dev.h_cptr2, dev.h_cidx2, dev.lev[l].ext2_first, dev.lev[l].ext2_count, dev.lev[l].ext2_countA);
    // border tiles of this level's fronts that carry unknowns of the parents' first 256 columns (a front's border lists its
    // parent's unknowns first, in the parent's own order)
    NdLevel& L = dev.lev[l];
    L.split_ta = 0;
    for (int k = 0; k < L.n; ++k) {
      const int n = order[L.first + k], p = hp.parent[n];
      if (p < 0) continue;
      int d = 0;
      for (int v : hp.strct[n]) if (hp.vnode[v] == p && hp.voff[v] < 256) d += NdHostPlan::vdim(v);
      L.split_ta = std::max(L.split_ta, (d + kTile - 1) / kTile);
    }
  }
  // flattened work / child lists of the extend-add kernel (NdDev::extw / extc / extc2)
  {
    auto flat = [&](const std::vector<int>& cidx, std::vector<int>& out) {
      for (int ch : cidx) {
        const unsigned long long mo = (unsigned long long)dev.h_ntab[2 * (size_t)ch];
        out.push_back((int)(unsigned)(mo & 0xffffffffull)); out.push_back((int)(unsigned)(mo >> 32));
        out.push_back((int)dev.h_ntab[2 * (size_t)ch + 1]); out.push_back(dev.h_rhs_node[ch]); out.push_back(dev.h_inv_off[ch]); out.push_back(0);
      }
    };
    flat(dev.h_cidx, dev.h_extc); flat(dev.h_cidx2, dev.h_extc2);
    const size_t nt = dev.h_ext.size() / 3;
    dev.h_extw.resize(8 * nt);
    for (size_t q = 0; q < nt; ++q) {
      const int node = dev.h_ext[3 * q];
      const std::vector<int>& cp = dev.h_ext_kind[q] ? dev.h_cptr2 : dev.h_cptr;
      const unsigned long long fo = (unsigned long long)dev.h_ntab[2 * (size_t)node];
      int* w = dev.h_extw.data() + 8 * q;
      w[0] = (int)(unsigned)(fo & 0xffffffffull); w[1] = (int)(unsigned)(fo >> 32); w[2] = (int)dev.h_ntab[2 * (size_t)node + 1]; w[3] = dev.h_rhs_node[node];
      w[4] = dev.h_ext[3 * q + 1]; w[5] = dev.h_ext[3 * q + 2]; w[6] = cp[node]; w[7] = cp[node + 1];
    }
    // Round 6, the same lists as RECORDS of kExtRec ints (k_nd_extend_rec): per tile ONE record {header | first contributing child | that child's row
    // maps for the tile's 64 rows and 64 columns}, found by the tile's index alone, and one overflow record per further contributing child. The kernel's
    // chain of dependent loads is record -> values instead of tile -> child entries -> row maps -> values; children that reach neither the tile's rows
    // nor its columns are not visited at all (their terms were zero: same sums, same order).
    dev.h_extr.assign((size_t)kExtRec * nt, -1);
    dev.ext_over = nt;
    for (size_t q = 0; q < nt; ++q) {
      const int node = dev.h_ext[3 * q], a = dev.h_ext[3 * q + 1], b = dev.h_ext[3 * q + 2];
      const std::vector<int>& cp = dev.h_ext_kind[q] ? dev.h_cptr2 : dev.h_cptr;
      const std::vector<int>& ci = dev.h_ext_kind[q] ? dev.h_cidx2 : dev.h_cidx;
      int nrec = 0;
      const size_t next = dev.h_extr.size() / kExtRec - nt;
      for (int k = cp[node]; k < cp[node + 1]; ++k) {
        const int ch = ci[k];
        const int* inv = dev.h_inv.data() + dev.h_inv_off[ch];
        int maps[128];
        bool anyr = false, anyc = false;
        for (int e = 0; e < 64; ++e) {
          const int r = 64 * a + e, c = 64 * b + e;
          maps[e] = r < ld[node] ? inv[r] : -1; maps[64 + e] = c < ld[node] ? inv[c] : -1;
          anyr = anyr || maps[e] >= 0; anyc = anyc || maps[64 + e] >= 0;
        }
        if (!anyr || !anyc) continue;
        if (nrec > 0) dev.h_extr.resize(dev.h_extr.size() + kExtRec, -1);
        int* r = nrec == 0 ? dev.h_extr.data() + (size_t)kExtRec * q : dev.h_extr.data() + dev.h_extr.size() - kExtRec;
        const unsigned long long mo = (unsigned long long)dev.h_ntab[2 * (size_t)ch];
        r[8] = (int)(unsigned)(mo & 0xffffffffull); r[9] = (int)(unsigned)(mo >> 32); r[10] = (int)dev.h_ntab[2 * (size_t)ch + 1]; r[11] = dev.h_rhs_node[ch];
        for (int e = 0; e < 128; ++e) r[16 + e] = maps[e];
        ++nrec;
      }
      int* w = dev.h_extr.data() + (size_t)kExtRec * q;
      for (int e = 0; e < 6; ++e) w[e] = dev.h_extw[8 * q + e];
      w[6] = nrec; w[7] = (int)next;
    }
  }
  // border x border 128-tiles that receive something from a child (NdDev::bb): only those are cleared per iteration and read by the first
  // panel's trailing update. A top front of a sharded solve is summed over the ranks as it stands: all its tiles count as contributed to.
  dev.h_bb_off.assign(nl, 0); dev.h_bb.clear();
  for (int i = 0; i < nl; ++i) {
    const int nI = dev.h_nI[i], lb = (ld[i] - nI + kTile - 1) / kTile;
    dev.h_bb_off[i] = (int)dev.h_bb.size();
    dev.h_bb.resize(dev.h_bb.size() + (size_t)lb * (lb + 1) / 2, is_top(order[i]) ? 1 : 0);
    if (is_top(order[i]) || lb == 0) continue;
    int* bb = dev.h_bb.data() + dev.h_bb_off[i];
    auto children = [&](const std::vector<int>& cptr, const std::vector<int>& cidx) {
      for (int q = cptr[i]; q < cptr[i + 1]; ++q) {
        const int* inv = dev.h_inv.data() + dev.h_inv_off[cidx[q]];
        std::vector<int> rows;
        for (int t = 0; t < lb; ++t) {
          bool any = false;
          for (int r = nI + kTile * t; r < std::min(ld[i], nI + kTile * t + kTile) && !any; ++r) any = inv[r] >= 0;
          if (any) rows.push_back(t);
        }
        for (int a : rows) for (int b : rows) if (b <= a) bb[a * (a + 1) / 2 + b] = 1;
      }
    };
    children(dev.h_cptr, dev.h_cidx);
    children(dev.h_cptr2, dev.h_cidx2);
  }
  {  // bottom levels whose backward substitution runs as one launch (k_panel.hip: k_bwd_tree): while a level's fronts hold at most two interior tiles
    int ls = 0;
    const int lmax = std::min(std::min(nlev, Hs), kBwdTreeMax);
    while (ls < lmax && dev.lev[ls].n > 0 && dev.lev[ls].n <= 65535 && (dev.lev[ls].own_max + kTile - 1) / kTile <= 2) ++ls;
    dev.tree_levels = ls >= 2 ? ls : 0;
    // the top levels, from the root down while the tile workgroups stay within 128 (k_bwd_tree64: a workgroup takes a whole CU) and every front has two
    // interior tiles or more somewhere in the level
    int lt = nlev, wgs = 0;
    while (lt > dev.tree_levels && nlev - lt < kBwdTreeMax) {
      const NdLevel& L = dev.lev[lt - 1];
      const int T = std::min(L.nI / kTile, (L.own_max + kTile - 1) / kTile);
      if (L.n <= 0 || T < 2 || wgs + L.n * T > 128) break;
      wgs += L.n * T; --lt;
    }
    dev.tree_top0 = nlev - lt >= 2 ? lt : nlev;
    dev.h_tree_fill.clear();
    for (int l = 0; l < nlev; ++l)
      if (l < dev.tree_levels || l >= dev.tree_top0)
        for (int i = dev.lev[l].first; i < dev.lev[l].first + dev.lev[l].n; ++i)
          for (int q = 0; q < dev.h_own_dims[i]; ++q) dev.h_tree_fill.push_back(dev.h_gidx[dev.h_own_g[i] + q]);
  }
  dev.active = true;
}

// ------------------------------------------------------------------------------------------------ device kernels
struct NdLevArgs {
  int first, n, nI, ntot;
  double* rhs;  // the level's right-hand sides: [n][2 ntot]
  const int *own_dims, *st_dims, *own_g, *st_g, *gidx, *inv_off, *inv;
};

// clear the tiles the factorisation will touch and put the identity on the interior padding rows. One workgroup per 128x128
// tile; dead tiles exit at once. Touched: the lower triangle over [interior tiles up to the last big panel that holds a real
// column | real border tiles] (a half-padded 256-column panel is read whole by the panel kernels), and the 2x2 diagonal
// tiles of the all-padding panels (k_potrf_panel factors every panel of every front of the batch).
// (ONE launch over the tiles of all levels — a level per launch was seven dependent host enqueues at the head of every linearisation,
//  with the chip idle behind them: blocks [blk0[l], blk0[l + 1]) belong to level l, tile (tr, tc) of its front number `fz`)
struct NdZeroArgs { int nlev; int first[24], n[24], nI[24], T[24]; long long blk0[25]; const int *own_dims, *st_dims, *bb_off, *bb; int first_top; };
// the extend-add stores the border x border tiles whole instead of k_nd_zero clearing them (COVGPU_STORE_BORDER=0: round 5's clearing; needs COVGPU_BETA0 on)
static bool nd_store_border() {
  static const bool v = (getenv("COVGPU_STORE_BORDER") == nullptr || atoi(getenv("COVGPU_STORE_BORDER")) != 0) && (getenv("COVGPU_BETA0") == nullptr || atoi(getenv("COVGPU_BETA0")) != 0);
  return v;
}
__global__ __launch_bounds__(256) void k_nd_zero(DevProblem P, NdZeroArgs z) {
  int l = 0;
  while (l + 1 < z.nlev && (long long)blockIdx.x >= z.blk0[l + 1]) ++l;
  const int T = z.T[l];
  const int q = (int)((long long)blockIdx.x - z.blk0[l]);
  const int fz = q / (T * T), rem = q - fz * T * T, tr = rem / T, tc = rem - tr * T;
  if (tc > tr) return;
  const int node = z.first[l] + fz;
  // (a front of at most 128 unknowns: nothing reads its second tile column below the panel's own 2x2 tiles — the panel factorisation, the
  //  substitutions and the rank updates all stop at the front's own last real 16-column block)
  const int own_n = z.own_dims[node];
  const int nIt = z.nI[l] / kTile, lo2 = own_n <= kTile ? 1 : 2 * ((own_n + 2 * kTile - 1) / (2 * kTile)), lb = (z.st_dims[node] + kTile - 1) / kTile;
  auto live = [&](int t) { return t < lo2 || (t >= nIt && t - nIt < lb); };
  const bool pad_diag = tr >= lo2 && tr < nIt && tc >= (tr & ~1);
  if (!((live(tr) && live(tc)) || pad_diag)) return;
  // a border x border tile no child adds into: the first panel's trailing update starts it from zero itself (GemmArgs::beta0); one that a child
  // does add into is written whole by the extend-add (round 6: store_border) — except in the replicated top fronts of a sharded solve, which are
  // summed over the ranks as they stand
  if (z.bb != nullptr && tc >= nIt && (node < z.first_top || !z.bb[z.bb_off[node] + (tr - nIt) * (tr - nIt + 1) / 2 + (tc - nIt)])) return;
  const size_t ld = (size_t)P.nd_ntab[2 * node + 1];
  double* M = P.nd_M + P.nd_ntab[2 * node] + (size_t)tr * kTile * ld + (size_t)tc * kTile;
  const int own = z.own_dims[node];
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
__global__ __launch_bounds__(256) void k_nd_assemble(DevProblem P, const int* __restrict__ rhs_off, double* __restrict__ x, const int* __restrict__ fill, int nfill,
                                                      unsigned long long empty) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // (the solution vector's entries of the levels whose backward substitution is ONE launch start a solve "empty": k_panel.hip, k_bwd_tree)
  if (t < nfill) reinterpret_cast<unsigned long long*>(x)[fill[t]] = empty;
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
    if (node >= 0) P.nd_rhs[rhs_off[node] + P.nd_voff[v] + rr] = P.bred[t];  // (node < 0: an unknown of another rank's subtree)
  }
}

// parent front += children's Schur complements, parent right-hand side += children's reduced right-hand sides, in child order.
// One workgroup per 64x64 tile of a host-built list (only tiles some child contributes to); a thread owns a 4x4 sub-grid.
// Round 5: tile and child descriptors flattened on the host (NdDev::extw / extc): tile -> child entry -> row map -> value, with the child entry two
// and the row maps one child ahead of the gather — the kernel is a chain of dependent loads on the critical path of every level transition.
__global__ __launch_bounds__(256) void k_nd_extend(DevProblem P, NdLevArgs a, const int* __restrict__ work, const int* __restrict__ child, int second_pass, int store_border, DevSignal sig) {
  // ("the chain's stream has finished the level below": published by this launch's first thread instead of a launch of its own — CholAux::publish_handle)
  if (sig.flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int4 w0 = reinterpret_cast<const int4*>(work)[2 * (size_t)blockIdx.x], w1 = reinterpret_cast<const int4*>(work)[2 * (size_t)blockIdx.x + 1];
  const size_t foff = (size_t)(unsigned)w0.x | ((size_t)(unsigned)w0.y << 32);
  const size_t ld = (size_t)w0.z;
  const int rhs_p = w0.w, tr = w1.x, tc = w1.y, kbeg = w1.z, kend = w1.w;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  double* F = P.nd_M + foff;
  const int nrow = (int)ld;
  int rr[4], cc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rr[i] = 64 * tr + ty + 16 * i; cc[i] = 64 * tc + tx + 16 * i; }
  struct Ch { size_t moff; int ldc, rhs, inv; };
  auto load_ch = [&](int k) {
    const int2* q = reinterpret_cast<const int2*>(child) + 3 * (size_t)k;
    const int2 a0 = q[0], a1 = q[1], a2 = q[2];
    Ch c; c.moff = (size_t)(unsigned)a0.x | ((size_t)(unsigned)a0.y << 32); c.ldc = a1.x; c.rhs = a1.y; c.inv = a2.x;
    return c;
  };
  Ch c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};   // entries of child k + 1 | k + 2
  if (kbeg < kend) c1 = load_ch(kbeg);
  if (kbeg + 1 < kend) c2 = load_ch(kbeg + 1);
  // what the front holds already: only columns of the front's OWN unknowns carry original entries (a border x border entry
  // belongs to an ancestor's front) — those are read up front, beside the index loads, the rest starts from zero
  double v[4][4], rv[4] = {0.0, 0.0, 0.0, 0.0};
  bool hit[4][4];
  // (second_pass: the top fronts of a sharded solve take their top children after the all-reduce — everything is read then)
  const bool own_cols = second_pass != 0 || 64 * tc < a.nI;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[i][j] = (own_cols && rr[i] < nrow && cc[j] <= rr[i]) ? F[(size_t)rr[i] * ld + cc[j]] : 0.0;
      hit[i][j] = false;
    }
  // children in child order (deterministic sums); the row maps of child k + 1 are fetched before the entries of child k are gathered
  int irn[4], icn[4];
  auto load_maps = [&](const Ch& c) {
    const int* inv = a.inv + c.inv;
#pragma unroll
    for (int i = 0; i < 4; ++i) { irn[i] = rr[i] < nrow ? inv[rr[i]] : -1; icn[i] = cc[i] < nrow ? inv[cc[i]] : -1; }
  };
  if (kbeg < kend) load_maps(c1);
  for (int k = kbeg; k < kend; ++k) {
    const Ch c = c1;
    c1 = c2;
    if (k + 2 < kend) c2 = load_ch(k + 2);
    const double* C = P.nd_M + c.moff;
    const size_t ldc = (size_t)c.ldc;
    int ir[4], ic[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ir[i] = irn[i]; ic[i] = icn[i]; }
    if (k + 1 < kend) load_maps(c1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ir[i] < 0) continue;
      if (tr == tc && tx == 0) rv[i] += P.nd_rhs[c.rhs + ir[i]];  // (diagonal tiles: lane tx == 0 of a row folds the right-hand side)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ic[j] < 0 || cc[j] > rr[i]) continue;
        v[i][j] += C[(size_t)(ir[i] > ic[j] ? ir[i] : ic[j]) * ldc + (ir[i] > ic[j] ? ic[j] : ir[i])];
        hit[i][j] = true;
      }
    }
  }
  // border x border tiles (no original entries, never cleared: k_nd_zero): EVERY entry of the lower part is stored — what no child reaches is zero
  const bool store_all = !own_cols && store_border != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (hit[i][j] || (store_all && rr[i] < nrow && cc[j] <= rr[i])) F[(size_t)rr[i] * ld + cc[j]] = v[i][j];
    if (rv[i] != 0.0) P.nd_rhs[rhs_p + rr[i]] += rv[i];
  }
}

// The same extend-add reading RECORDS (nd_tables: NdDev::extr): the level transitions sit on the serial chain with a dozen workgroups each, and a
// workgroup of k_nd_extend is a chain of six dependent loads under a chip full of trailing updates (tile -> child entries -> row maps -> values, a
// child at a time, then the right-hand side's read-modify-write) — 35-50 us for ten tiles. Here: record (header, first child and its maps: one
// address, known from the tile index) -> values (and the overflow records of further children) -> store. Bit-identical to k_nd_extend.
__global__ __launch_bounds__(256) void k_nd_extend_rec(DevProblem P, NdLevArgs a, const int* __restrict__ recs, const int* __restrict__ over, int second_pass, int store_border, DevSignal sig) {
  if (sig.flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int* rec = recs + (size_t)kExtRec * blockIdx.x;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int4 w0 = reinterpret_cast<const int4*>(rec)[0], w1 = reinterpret_cast<const int4*>(rec)[1];
  int4 chn = reinterpret_cast<const int4*>(rec)[2];
  int irn[4], icn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { irn[i] = rec[16 + ty + 16 * i]; icn[i] = rec[80 + tx + 16 * i]; }
  const size_t foff = (size_t)(unsigned)w0.x | ((size_t)(unsigned)w0.y << 32);
  const size_t ld = (size_t)w0.z;
  const int rhs_p = w0.w, tr = w1.x, tc = w1.y, nrec = w1.z, next = w1.w;
  double* F = P.nd_M + foff;
  const int nrow = (int)ld;
  int rr[4], cc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rr[i] = 64 * tr + ty + 16 * i; cc[i] = 64 * tc + tx + 16 * i; }
  double v[4][4], rv[4] = {0.0, 0.0, 0.0, 0.0};
  bool hit[4][4];
  const bool own_cols = second_pass != 0 || 64 * tc < a.nI;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[i][j] = (own_cols && rr[i] < nrow && cc[j] <= rr[i]) ? F[(size_t)rr[i] * ld + cc[j]] : 0.0;
      hit[i][j] = false;
    }
  for (int k = 0; k < nrec; ++k) {
    const double* C = P.nd_M + ((size_t)(unsigned)chn.x | ((size_t)(unsigned)chn.y << 32));
    const size_t ldc = (size_t)chn.z;
    const int crhs = chn.w;
    int ir[4], ic[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ir[i] = irn[i]; ic[i] = icn[i]; }
    if (k + 1 < nrec) {   // the next child's record is in flight while this one's entries are gathered
      const int* r2 = over + (size_t)kExtRec * (next + k);
      chn = reinterpret_cast<const int4*>(r2)[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) { irn[i] = r2[16 + ty + 16 * i]; icn[i] = r2[80 + tx + 16 * i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ir[i] < 0) continue;
      if (tr == tc && tx == 0) rv[i] += P.nd_rhs[crhs + ir[i]];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ic[j] < 0 || cc[j] > rr[i]) continue;
        v[i][j] += C[(size_t)(ir[i] > ic[j] ? ir[i] : ic[j]) * ldc + (ir[i] > ic[j] ? ic[j] : ir[i])];
        hit[i][j] = true;
      }
    }
  }
  const bool store_all = !own_cols && store_border != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (hit[i][j] || (store_all && rr[i] < nrow && cc[j] <= rr[i])) F[(size_t)rr[i] * ld + cc[j]] = v[i][j];
    if (rv[i] != 0.0) P.nd_rhs[rhs_p + rr[i]] += rv[i];
  }
}

// block inverses of the padding columns are the identity once and for all (the panel kernel never writes them: DenseBatch::own_max)
__global__ __launch_bounds__(256) void k_nd_linv_init(double* __restrict__ Linv, size_t n) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) { const int e = (int)(q & 255); Linv[q] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0; }
}

// sharded solve: [grad | hdiag] of the top unknowns <-> the tail of the all-reduced range (dir 0: pack, 1: unpack)
__global__ __launch_bounds__(256) void k_nd_gh(DevProblem P, const int* __restrict__ top_g, int ntop, double* __restrict__ buf, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntop) return;
  const int q = top_g[i];
  if (dir == 0) { buf[i] = P.grad[q]; buf[ntop + i] = P.hdiag[q]; }
  else { P.grad[q] = buf[i]; P.hdiag[q] = buf[ntop + i]; }
}
// sharded solve: the live lower tiles of the top fronts <-> the packed exchange buffer (dir 0: pack, 1: unpack). One workgroup per tile.
__global__ __launch_bounds__(256) void k_nd_top_pack(DevProblem P, const int* __restrict__ tiles, double* __restrict__ pack, int dir) {
  const int node = tiles[3 * blockIdx.x], tr = tiles[3 * blockIdx.x + 1], tc = tiles[3 * blockIdx.x + 2];
  const size_t ld = (size_t)P.nd_ntab[2 * node + 1];
  double* M = P.nd_M + P.nd_ntab[2 * node] + (size_t)tr * kTile * ld + (size_t)tc * kTile;
  double* B = pack + (size_t)blockIdx.x * kTile * kTile;
  for (int e = threadIdx.x; e < kTile * kTile / 2; e += 256) {
    const int r = e / (kTile / 2), c2 = 2 * (e - r * (kTile / 2));
    double2* m = reinterpret_cast<double2*>(M + (size_t)r * ld + c2);
    double2* b = reinterpret_cast<double2*>(B + (size_t)r * kTile + c2);
    if (dir == 0) *b = *m; else *m = *b;
  }
}
// sharded solve: trust-region damping of the top unknowns, applied to the all-reduced top fronts with the all-reduced
// diag(J^T J) (k_finalize_diag leaves them out: every rank holds only its part of their rows before the exchange)
__global__ __launch_bounds__(256) void k_nd_top_damp(DevProblem P, const int* __restrict__ top_var, const int* __restrict__ top_r,
                                                      const int* __restrict__ top_g, int ntop, const int* __restrict__ rhs_off, double mu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntop) return;
  const int v = top_var[i], r = top_r[i];
  const double h = P.hdiag[top_g[i]];
  double* d = nd_entry(P, v, v, r, r);
  if (h == 0.0) { *d = 1.0; P.nd_rhs[rhs_off[P.nd_vnode[v]] + P.nd_voff[v] + r] = 0.0; }
  else { const double c = covdev::clamp_diag(h); *d += mu * c * c; }
}

// ------------------------------------------------------------------------------------------------ launchers
static NdLevArgs lev_args(const DevProblem& P, const NdDev& nd, int l) {
  const NdLevel& L = nd.lev[l];
  return NdLevArgs{L.first, L.n, L.nI, L.ntot, P.nd_rhs + L.rhs_off, nd.own_dims, nd.st_dims, nd.own_g, nd.st_g, nd.gidx, nd.inv_off, nd.inv};
}

void launch_nd_init(const DevProblem& P, const NdDev& nd, hipStream_t st) {
  if (nd.linv_elems) hipLaunchKernelGGL(k_nd_linv_init, dim3((unsigned)((nd.linv_elems + 255) / 256)), dim3(256), 0, st, P.nd_Linv, nd.linv_elems);
}

void launch_nd_zero(const DevProblem& P, const NdDev& nd, hipStream_t st) {
  NdZeroArgs z;
  static const bool beta0 = getenv("COVGPU_BETA0") == nullptr || atoi(getenv("COVGPU_BETA0")) != 0;   // (A/B switch; k_front.hip's batch() reads the same)
  z.nlev = 0; z.blk0[0] = 0; z.own_dims = nd.own_dims; z.st_dims = nd.st_dims; z.bb_off = nd.bb_off; z.bb = beta0 ? nd.bb : nullptr;
  // nodes are numbered in level order: the first node of the first top level (sharded solve) — below it the extend-add stores the border tiles
  z.first_top = !nd_store_border() ? 0 : (nd.top_lev0 < (int)nd.lev.size() ? nd.lev[nd.top_lev0].first : nd.nnodes);
  auto flush = [&] {
    if (z.nlev > 0 && z.blk0[z.nlev] > 0) hipLaunchKernelGGL(k_nd_zero, dim3((unsigned)z.blk0[z.nlev]), dim3(256), 0, st, P, z);
    z.nlev = 0; z.blk0[0] = 0;
  };
  for (size_t l = 0; l < nd.lev.size(); ++l) {
    const NdLevel& L = nd.lev[l];
    if (L.n == 0) continue;
    const int T = L.ntot / kTile;
    const long long nb = (long long)T * T * L.n;
    if (z.nlev == 24 || z.blk0[z.nlev] + nb > 0x7fffff00ll) flush();   // (more than 24 levels / 2^31 blocks: another launch)
    z.first[z.nlev] = L.first; z.n[z.nlev] = L.n; z.nI[z.nlev] = L.nI; z.T[z.nlev] = T;
    z.blk0[z.nlev + 1] = z.blk0[z.nlev] + nb;
    ++z.nlev;
  }
  flush();
}

bool launch_nd_solve(const DevProblem& P, NdDev& nd, double* dst, double mu, hipStream_t st, CholAux& ax) {
  const int nlev = (int)nd.lev.size(), ltop = nd.top_lev0;
  static const bool lookahead = getenv("COVGPU_ND_LOOKAHEAD") == nullptr || atoi(getenv("COVGPU_ND_LOOKAHEAD")) != 0;
  ax.init();
  {  // scratch of the one-launch backward substitution (k_bwd_front): [fronts][interior tiles <= 4][256-row chunks of the border][128]
    size_t need = 0;
    for (const NdLevel& L : nd.lev)
      need = std::max(need, (size_t)L.n * std::min(bwd_front_max_tiles(), (L.own_max + kTile - 1) / kTile) * std::max(1, (L.ntot - L.nI + 255) / 256) * kTile);
    if (need > ax.bwd_scr_elems) {
      if (ax.bwd_scr) (void)hipFree(ax.bwd_scr);
      ax.bwd_scr = nullptr; ax.bwd_scr_elems = 0;
      if (hipMalloc((void**)&ax.bwd_scr, need * sizeof(double)) != hipSuccess) { ax.bwd_scr = nullptr; return false; }   // (nothing enqueued yet)
      ax.bwd_scr_elems = need;
    }
  }
  if (!ax.pipe_broken && ax.gate_dead != nullptr && ax.gate_dead_h != nullptr) {  // scratch of the pipelined backward substitution (k_bwd_pipe): [fronts][interior tiles][chunks + 1][128], filled once
    size_t need = 0;
    for (const NdLevel& L : nd.lev) {
      const int T = std::min(L.nI / kTile, (L.own_max + kTile - 1) / kTile);
      if (L.n > 0 && T >= bwd_pipe_min_tiles()) need = std::max(need, (size_t)L.n * T * ((L.ntot - L.nI + kPipeChunk - 1) / kPipeChunk + 1) * kTile);
    }
    {
      size_t tree = 0, tree2 = 0;
      for (int l = 0; l < (int)nd.lev.size(); ++l) {
        const NdLevel& L = nd.lev[l];
        const size_t e = (size_t)L.n * std::min(L.nI / kTile, (L.own_max + kTile - 1) / kTile) * ((L.ntot - L.nI + kPipeChunk - 1) / kPipeChunk + 1) * kTile;
        if (l < nd.tree_levels) tree += e;
        if (l >= nd.tree_top0) tree2 += e;
      }
      need = std::max(need, std::max(tree, tree2));
    }
    if (need > ax.bwd_pipe_elems) {
      if (ax.bwd_pipe) { (void)hipDeviceSynchronize(); (void)hipFree(ax.bwd_pipe); }
      ax.bwd_pipe = nullptr; ax.bwd_pipe_elems = 0;
      if (hipMalloc((void**)&ax.bwd_pipe, need * sizeof(double)) == hipSuccess) {
        ax.bwd_pipe_elems = need;
        launch_pipe_fill(ax.bwd_pipe, need, st);   // (stream-ordered before the first use)
      } else ax.bwd_pipe = nullptr;                // (launch per tile)
    }
  }
  static const bool fused_bwd = getenv("COVGPU_ND_BWD_FUSED") == nullptr || atoi(getenv("COVGPU_ND_BWD_FUSED")) != 0;
  static const bool tree_env = getenv("COVGPU_BWD_TREE") == nullptr || atoi(getenv("COVGPU_BWD_TREE")) != 0;
  // the bottom levels' backward substitution as one launch (k_bwd_tree): needs the pipeline's scratch and the give-up word
  const bool tree_ok = tree_env && fused_bwd && nd.tree_fill != nullptr && !ax.pipe_broken && ax.bwd_pipe != nullptr && ax.gate_dead != nullptr && bwd_pipe_min_tiles() <= 2;
  static const bool tree_top_env = getenv("COVGPU_BWD_TREE_TOP") == nullptr || atoi(getenv("COVGPU_BWD_TREE_TOP")) != 0;
  const bool tree_on = tree_ok && nd.tree_levels >= 2;
  const bool tree_top_on = tree_ok && tree_top_env && nd.tree_top0 < (int)nd.lev.size();
  ax.mark(st, -1);
  // (the right-hand sides were cleared on the head stream of the build, beside the fronts: solver.hip enqueue_build)
  {
    const int nfill = (tree_on || tree_top_on) ? (int)nd.h_tree_fill.size() : 0;
    const int cnt = std::max(std::max(P.vi ? 324 * P.K : 0, P.n), nfill);
    hipLaunchKernelGGL(k_nd_assemble, dim3((cnt + 255) / 256), dim3(256), 0, st, P, (const int*)nd.rhs_node, dst, (const int*)nd.tree_fill, nfill, pipe_empty_word());
  }
  ax.mark(st, -2);
  auto batch = [&](int l) {
    const NdLevel& L = nd.lev[l];
    DenseBatch bt;
    bt.n = L.n; bt.sM = 0; bt.sL = (size_t)L.nI * kTile; bt.sR = (size_t)2 * L.ntot;
    bt.live = L.live; bt.tI = L.nI / kTile; bt.live_h = L.live_h.data();
    static const bool beta0 = getenv("COVGPU_BETA0") == nullptr || atoi(getenv("COVGPU_BETA0")) != 0;
    if (beta0 && nd.bb != nullptr) { bt.beta0_off = nd.bb_off + L.first; bt.beta0 = nd.bb; }
    static const bool plists = getenv("COVGPU_POTRF_LISTS") == nullptr || atoi(getenv("COVGPU_POTRF_LISTS")) != 0;
    if (plists && L.plist != nullptr) { bt.plist = L.plist; bt.pbig_h = L.pbig.data(); bt.psmall_h = L.psmall.data(); }
    bt.tab = P.nd_ntab + 2 * (size_t)L.first; bt.tri_slot = l; bt.own_max = L.own_max; bt.own_dims = nd.own_dims + L.first; bt.own_dims_h = nd.h_own_dims.data() + L.first;
    return bt;
  };
  // part: 0 all tiles | 1 the tiles of the fronts' first 256 rows | 2 the others
  auto extend = [&](int l, bool top_children, int part, hipStream_t s2, DevSignal sig = DevSignal()) -> bool {
    const NdLevel& L = nd.lev[l];
    int first = top_children ? L.ext2_first : L.ext_first, count = top_children ? L.ext2_count : L.ext_count;
    const int countA = top_children ? L.ext2_countA : L.ext_countA;
    if (part == 1) count = countA;
    if (part == 2) { first += countA; count -= countA; }
    static const bool records = getenv("COVGPU_EXT_RECORDS") == nullptr || atoi(getenv("COVGPU_EXT_RECORDS")) != 0;
    if (count > 0 && records && nd.extr != nullptr)
      hipLaunchKernelGGL(k_nd_extend_rec, dim3(count), dim3(256), 0, s2, P, lev_args(P, nd, l), (const int*)(nd.extr + (size_t)kExtRec * first),
                         (const int*)(nd.extr + (size_t)kExtRec * nd.ext_over), top_children ? 1 : 0, (!top_children && l < ltop && nd_store_border()) ? 1 : 0, sig);
    else if (count > 0)
      hipLaunchKernelGGL(k_nd_extend, dim3(count), dim3(256), 0, s2, P, lev_args(P, nd, l), (const int*)(nd.extw + 8 * (size_t)first),
                         (const int*)(top_children ? nd.extc2 : nd.extc), top_children ? 1 : 0, (!top_children && l < ltop && nd_store_border()) ? 1 : 0, sig);
    return count > 0;
  };
  // split: the trailing update of this level's last panel is split for the look-ahead into the next level (DenseBatch::split_ta)
  auto factor = [&](int l, bool split, hipEvent_t pre_trsm) {
    const NdLevel& L = nd.lev[l];
    ax.mark(st, -3);
    if (L.n > 0) {
      DenseBatch bt = batch(l);
      bt.split_ta = split ? L.split_ta : 0; bt.pre_trsm = pre_trsm;
      dense_cholesky_solve_raw(P.nd_M, P.nd_rhs + L.rhs_off, P.nd_Linv + L.linv_off, P.flag, L.ntot, st, ax, L.nI / kTile, false, bt);
    }
    ax.mark(st, -4);
  };
  // Levels [l0, l1) one after the other, with a look-ahead across the level boundary: of level l's last trailing update only the
  // tiles its parents' first panel receives run on the chain's stream, followed by that part of the extend-add and the parents'
  // first panel; the rest of the update and of the extend-add run beside it on the bulk stream (ax.aux).
  auto run_levels = [&](int l0, int l1, bool top_children) {
    bool prev_split = false;
    for (int l = l0; l < l1; ++l) {
      hipEvent_t pre = nullptr;
      if (prev_split) {
        // (the bulk stream already follows the level below through its own update; this also covers a batch that declined the split. Round 6: the
        //  record rides with the first half of the extend-add — published by its first thread — when that launch exists)
        DevSignal sig = ax.publish_handle(ax.ev_xa, st, 5);
        const int cntA = top_children ? nd.lev[l].ext2_countA : nd.lev[l].ext_countA;
        if (sig.flag == nullptr || cntA <= 0) { if (sig.flag != nullptr) ax.record_handle(sig, st); else ax.record(ax.ev_xa, st, 5); sig = DevSignal(); }
        extend(l, top_children, 1, st, sig);
        ax.wait(ax.aux, ax.ev_xa);
        extend(l, top_children, 2, ax.aux);
        ax.record(ax.ev_xb, ax.aux, 6);
        pre = ax.ev_xb;
      } else extend(l, top_children, 0, st);
      const NdLevel& L = nd.lev[l];
      const bool split = lookahead && l + 1 < l1 && L.n > 0 && nd.lev[l + 1].n > 0 && L.split_ta > 0;
      factor(l, split, pre);
      prev_split = split;
    }
  };
  // ---- the subtrees of this rank (single GPU: the whole tree), level by level
  run_levels(0, ltop, false);
  if (ltop < nlev) {
    // ---- agent-sharded solve: the top fronts so far hold THIS rank's residuals and subtrees only. One
    //      all-reduce over [top fronts | their right-hand sides | grad, hdiag of the top unknowns] (contiguous), then the
    //      damping of the top unknowns; from here on every rank runs the top of the tree redundantly, with no further exchange.
    for (int l = ltop; l < nlev; ++l) extend(l, false, 0, st);
    double* gh = P.nd_rhs + nd.gh_off;
    if (nd.ntop > 0) hipLaunchKernelGGL(k_nd_gh, dim3((nd.ntop + 255) / 256), dim3(256), 0, st, P, (const int*)nd.top_g, nd.ntop, gh, 0);
    if (ax.reduce != nullptr) {
      // ONE all-reduce of [live lower tiles of the top fronts, packed | top right-hand sides | grad, hdiag of the top unknowns]: the
      // fronts are stored as full squares, of which the exchange needs half (33.6 -> 17.9 MB for the 5-agent map's root)
      const size_t ntile = (size_t)nd.n_top_tiles * kTile * kTile, nvec = nd.rhs_top + 2 * (size_t)nd.ntop;
      if (nd.top_pack != nullptr) {
        if (nd.n_top_tiles > 0) hipLaunchKernelGGL(k_nd_top_pack, dim3(nd.n_top_tiles), dim3(256), 0, st, P, (const int*)nd.top_tiles, nd.top_pack, 0);
        if (nvec > 0) (void)hipMemcpyAsync(nd.top_pack + ntile, P.nd_rhs, nvec * sizeof(double), hipMemcpyDeviceToDevice, st);
        ax.reduce(ax.reduce_ctx, nd.top_pack, ntile + nvec, 0, st);
        if (nd.n_top_tiles > 0) hipLaunchKernelGGL(k_nd_top_pack, dim3(nd.n_top_tiles), dim3(256), 0, st, P, (const int*)nd.top_tiles, nd.top_pack, 1);
        if (nvec > 0) (void)hipMemcpyAsync(P.nd_rhs, nd.top_pack + ntile, nvec * sizeof(double), hipMemcpyDeviceToDevice, st);
      } else ax.reduce(ax.reduce_ctx, P.nd_M + nd.M_sub, (nd.M_elems - nd.M_sub) + nd.rhs_top + 2 * (size_t)nd.ntop, 0, st);   // (whole squares: COVGPU_SHARD_PACK=0)
    }
    if (nd.ntop > 0) {
      hipLaunchKernelGGL(k_nd_gh, dim3((nd.ntop + 255) / 256), dim3(256), 0, st, P, (const int*)nd.top_g, nd.ntop, gh, 1);
      hipLaunchKernelGGL(k_nd_top_damp, dim3((nd.ntop + 255) / 256), dim3(256), 0, st, P, (const int*)nd.top_var, (const int*)nd.top_r, (const int*)nd.top_g, nd.ntop,
                         (const int*)nd.rhs_node, mu);
    }
    run_levels(ltop, nlev, true);
  }
  auto tree_launch = [&](int l_hi, int l_lo, bool form64) {   // levels l_hi .. l_lo (downwards) in one launch
    BwdTreeLevel tl[kBwdTreeMax];
    int nq = 0;
    for (int l = l_hi; l >= l_lo; --l) {
      const NdLevel& L = nd.lev[l];
      BwdTreeLevel& t = tl[nq++];
      t.nbt = L.n; t.T = std::min(L.nI / kTile, (L.own_max + kTile - 1) / kTile); t.nchunk = (L.ntot - L.nI + kPipeChunk - 1) / kPipeChunk; t.tI = L.nI / kTile; t.first = L.first;
      t.y = P.nd_rhs + L.rhs_off + L.ntot; t.bsR = (size_t)2 * L.ntot; t.Dinv = P.nd_Linv + L.linv_off; t.bsL = (size_t)L.nI * kTile;
      t.btab = P.nd_ntab + 2 * (size_t)L.first; t.live = L.live; t.scr_off = t.xpub_off = 0;
    }
    BwdXfer xf;
    xf.gidx = nd.gidx; xf.own_g = nd.own_g; xf.st_g = nd.st_g; xf.own_dims = nd.own_dims; xf.st_dims = nd.st_dims; xf.x = dst; xf.first = 0;
    launch_bwd_tree(P.nd_M, tl, nq, xf, ax.bwd_pipe, ax.gate_dead, ax.gate_dead_h, ax.gate_timeout_s, st, form64);
    ax.mark(st, -5);
  };
  if (tree_top_on) tree_launch(nlev - 1, nd.tree_top0, true);
  for (int l = (tree_top_on ? nd.tree_top0 : nlev) - 1; l >= (tree_on ? nd.tree_levels : 0); --l) {
    // top-down: the ancestors' unknowns are read from `dst` by the first launch of the level, the fronts' own unknowns are
    // written there by its last (BwdXfer)
    const NdLevel& L = nd.lev[l];
    if (L.n == 0) continue;
    DenseBatch bt = batch(l);
    bt.xfer.gidx = nd.gidx; bt.xfer.own_g = nd.own_g; bt.xfer.st_g = nd.st_g; bt.xfer.own_dims = nd.own_dims; bt.xfer.st_dims = nd.st_dims;
    bt.xfer.x = dst; bt.xfer.first = L.first;
    bt.bwd_cnt = fused_bwd ? ax.bwd_cnt : nullptr; bt.bwd_scr = ax.bwd_scr;
    if (fused_bwd && !ax.pipe_broken && ax.bwd_pipe != nullptr) { bt.bwd_pipe = ax.bwd_pipe; bt.pipe_dead = ax.gate_dead; bt.pipe_dead_h = ax.gate_dead_h; bt.pipe_timeout_s = ax.gate_timeout_s; }
    dense_backward_solve(P.nd_M, P.nd_rhs + L.rhs_off, P.nd_Linv + L.linv_off, L.ntot, st, L.nI / kTile, L.ntot / kTile, bt);
    ax.mark(st, -5);
  }
  if (tree_on) tree_launch(nd.tree_levels - 1, 0, false);   // levels tree_levels - 1 .. 0 in one launch
  ax.mark(st, -6);
  return true;
}

}  // namespace covgpu
