// nd_plan.hip — host-only: nested-dissection tree + symbolic factorisation of the reduced camera system (nd_plan.hpp).
// No device code here; the file is a .hip only so that the one Makefile rule builds it.
#include "nd_plan.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <queue>
#include <thread>
#include <tuple>

namespace covgpu {
namespace {

struct Builder {
  const int K;
  const bool vi;
  const int leaf_dims;
  const std::vector<std::vector<int>>& adj;
  const std::vector<int>& chain_of;  // by position
  NdHostPlan& out;
  std::vector<int> reg, deg;
  std::vector<int> side;
  std::vector<char> inS;
  int regid = 0;
  int top_mode = 0;   // several agents in a region: 0 ONE cover of all cross-agent couplings | 1 two groups of agents, recursively
  double group_frac = 3.0;   // top_mode 1: the lighter group holds at least 1 / group_frac of the region's unknowns

  Builder(int K_, bool vi_, int leaf, const std::vector<std::vector<int>>& adj_, const std::vector<int>& chain_of_, NdHostPlan& o)
      : K(K_), vi(vi_), leaf_dims(leaf), adj(adj_), chain_of(chain_of_), out(o), reg(2 * (size_t)K_, -1), deg(2 * (size_t)K_, 0),
        side(2 * (size_t)K_, 0), inS(2 * (size_t)K_, 0) {}

  int new_node(std::vector<int>& own, int parent) {
    // speed-bias blocks first (inside a leaf they form the block-tridiagonal part that fills least), then poses, by position
    std::sort(own.begin(), own.end(), [](int a, int b) { return std::make_pair(1 - (a & 1), a >> 1) < std::make_pair(1 - (b & 1), b >> 1); });
    const int n = out.nnodes++;
    out.parent.push_back(parent);
    out.depth.push_back(parent < 0 ? 0 : out.depth[parent] + 1);
    out.child.emplace_back();
    if (parent >= 0) out.child[parent].push_back(n);
    int off = 0, ord = 0;
    for (int v : own) { out.vnode[v] = n; out.voff[v] = off; out.vord[v] = ord++; off += NdHostPlan::vdim(v); }
    out.own_dims.push_back(off);
    out.own.push_back(own);
    return n;
  }

  // A region that is not cut further becomes TWO fronts: its pose blocks (parent) and, below them, its speed-bias blocks.
  // Speed-bias blocks couple to their own and the neighbouring keyframes' poses and speed-bias blocks only — never to the far
  // border a trajectory segment carries (separators, shared keyframes of other agents). In one front [V | P | border] the
  // dense kernels would run the V columns through the whole border (zeros): 60 % of a leaf's columns. As a child front the V
  // chain is a small system of its own and the pose front above it is a single 256-column panel.
  void leaf(std::vector<int>& vars, int parent) {
    std::vector<int> P, V;
    for (int v : vars) ((v & 1) ? V : P).push_back(v);
    // The speed-bias chain is cut into segments of at most 14 blocks (126 unknowns: half a panel — the bottom level's panel
    // factorisation, substitutions and rank update all scale with the widest front; 10 / 14 / 21 / 28 blocks measured, 14 best):
    // a cut block joins the pose front — it is the separator of its two neighbours.
    constexpr int vseg = 14;
    if (V.empty() || (P.empty() && (int)V.size() <= vseg) || leaf_dims >= (1 << 29)) { new_node(vars, parent); return; }  // (COVGPU_GBA_DENSE: one front = the dense system)
    std::sort(V.begin(), V.end());
    const int nV = (int)V.size(), pd = 6 * (int)P.size();
    int s = (nV + 1 + vseg) / (vseg + 1);
    while (s > 1 && pd <= 256 && pd + 9 * (s - 1) > 256) --s;   // keep the pose front a single panel
    std::vector<std::vector<int>> seg(s);
    {
      const int rest = nV - (s - 1), base = rest / s, extra = rest % s;
      int q = 0;
      for (int k = 0; k < s; ++k) {
        for (int i = 0; i < base + (k < extra ? 1 : 0); ++i) seg[k].push_back(V[q++]);
        if (k + 1 < s) P.push_back(V[q++]);
      }
    }
    const int np = new_node(P, parent);
    for (auto& sg : seg) if (!sg.empty()) new_node(sg, np);
  }

  void build(std::vector<int>& vars, int parent) {
    if (vars.empty()) return;
    int dims = 0;
    for (int v : vars) dims += NdHostPlan::vdim(v);
    if (dims <= leaf_dims) { leaf(vars, parent); return; }
    // ---- parts of the cut. Several agents: one part per agent — ONE cover of all cross-agent couplings (loop-closure
    //      zones are hot spots where several agents meet: a keyframe there covers links to all of them at once; pairwise
    //      agent cuts needed 1.5x more separator unknowns on the 5-agent map). One agent: two halves of its time axis.
    std::vector<int> chains;
    for (int v : vars) chains.push_back(chain_of[v >> 1]);
    std::sort(chains.begin(), chains.end());
    chains.erase(std::unique(chains.begin(), chains.end()), chains.end());
    int nparts = 2;
    if (chains.size() >= 3 && top_mode == 1) {
      // Two GROUPS of agents (round 5): the cover of the couplings that cross between the groups; each group is cut again below it.
      // One cover of ALL cross-agent couplings (top_mode 0) puts every loop-closure zone of the map into ONE front — on the 5-agent
      // map with the ground-truth orientations read correctly 621 keyframes = 15 serial panels and 60 % of the flops.
      const int nc = (int)chains.size();
      constexpr int kBruteChains = 14;   // 2^13 bipartitions x nc per step: a millisecond
      std::vector<char> side_best;
      std::vector<long long> W((size_t)nc * nc, 0), size(nc, 0);
      ++regid;
      for (int v : vars) { reg[v] = regid; side[v] = (int)(std::lower_bound(chains.begin(), chains.end(), chain_of[v >> 1]) - chains.begin()); size[side[v]] += 1; }
      for (int v : vars) for (int w : adj[v]) if (reg[w] == regid && side[w] != side[v]) W[(size_t)side[v] * nc + side[w]] += 1;
      long long total = 0;
      for (int c = 0; c < nc; ++c) total += size[c];
      // Symmetric pair weights once: the crossing weight of a bipartition is then updated per moved chain (O(nc)) instead of recomputed (O(nc^2)).
      std::vector<long long> Ws((size_t)nc * nc, 0);
      for (int a = 0; a < nc; ++a) for (int b = 0; b < nc; ++b) Ws[(size_t)a * nc + b] = W[(size_t)a * nc + b] + W[(size_t)b * nc + a];
      auto balanced = [&](long long s1, double frac) { return (double)std::min(s1, total - s1) * frac >= (double)total; };
      uint64_t best = 0; long long best_w = -1;
      if (nc <= kBruteChains) {
        // all bipartitions with chain 0 on side 0, walked in Gray-code order (one chain changes side per step): lightest crossing weight among
        // those with at least 1 / group_frac of the unknowns on the lighter side (half of that, a quarter, ... if none qualifies). Ties: the
        // numerically smallest mask, as the plain enumeration of round 5 chose.
        for (double frac = group_frac; frac <= 48.0 && best_w < 0; frac *= 2.0) {
          uint64_t g = 0; long long s1 = 0, w = 0;
          const uint64_t nm = (uint64_t)1 << (nc - 1);
          for (uint64_t m = 1; m < nm; ++m) {
            const int c = 1 + __builtin_ctzll(m);   // the chain that changes side at this step (chain 0 never does)
            const bool to1 = !(g >> c & 1);
            long long d = 0;   // change of the crossing weight: links to the side it leaves start to cross, links to the side it joins stop
            for (int b = 0; b < nc; ++b) if (b != c) d += ((g >> b & 1) == (uint64_t)to1 ? -1 : 1) * Ws[(size_t)c * nc + b];
            w += d; g ^= (uint64_t)1 << c; s1 += to1 ? size[c] : -size[c];
            if (!balanced(s1, frac)) continue;
            if (best_w < 0 || w < best_w || (w == best_w && g < best)) { best_w = w; best = g; }
          }
        }
      } else {
        // Many chains (every keyframe without an IMU predecessor starts one: a map of short tracking sessions has dozens): 2^(nc-1) bipartitions are
        // out of reach (26 chains took 49 s) — a deterministic Fiduccia-Mattheyses-style local search instead: from a few fixed starts
        // (chains dealt by size to the lighter side; contiguous halves of the chain order) move the single chain with the best gain that keeps the
        // balance bound, until no move gains. O(nc^2) per pass.
        for (double frac = group_frac; frac <= 48.0 && best_w < 0; frac *= 2.0)
          for (int start = 0; start < 3; ++start) {
            std::vector<char> sd(nc, 0);
            if (start == 0) {
              std::vector<int> ord(nc);
              for (int c = 0; c < nc; ++c) ord[c] = c;
              std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return size[a] > size[b]; });
              long long l0 = 0, l1 = 0;
              for (int c : ord) { if (l1 < l0) { sd[c] = 1; l1 += size[c]; } else l0 += size[c]; }
            } else {
              long long acc = 0;
              for (int c = 0; c < nc; ++c) { const int cc = start == 1 ? c : nc - 1 - c; sd[cc] = acc * 2 >= total ? 1 : 0; acc += size[cc]; }
            }
            long long s1 = 0, w = 0;
            for (int c = 0; c < nc; ++c) if (sd[c]) s1 += size[c];
            for (int a = 0; a < nc; ++a) for (int b = a + 1; b < nc; ++b) if (sd[a] != sd[b]) w += Ws[(size_t)a * nc + b];
            for (int pass = 0; pass < 4 * nc; ++pass) {
              int pick = -1; long long pick_d = 0;
              for (int c = 0; c < nc; ++c) {
                const long long ns1 = s1 + (sd[c] ? -size[c] : size[c]);
                if (!balanced(ns1, frac) && balanced(s1, frac)) continue;   // (an unbalanced start may move towards balance)
                long long d = 0;
                for (int b = 0; b < nc; ++b) if (b != c) d += (sd[b] == sd[c] ? 1 : -1) * Ws[(size_t)c * nc + b];
                if (!balanced(s1, frac)) { if (std::min(ns1, total - ns1) <= std::min(s1, total - s1)) continue; d = -1 - std::min(ns1, total - ns1); }   // repair first
                if (d < pick_d) { pick_d = d; pick = c; }
              }
              if (pick < 0) break;
              if (balanced(s1, frac)) w += pick_d;
              else { long long d = 0; for (int b = 0; b < nc; ++b) if (b != pick) d += (sd[b] == sd[pick] ? 1 : -1) * Ws[(size_t)pick * nc + b]; w += d; }
              s1 += sd[pick] ? -size[pick] : size[pick]; sd[pick] ^= 1;
            }
            if (!balanced(s1, frac) || s1 == 0 || s1 == total) continue;
            if (best_w < 0 || w < best_w) { best_w = w; side_best.assign(sd.begin(), sd.end()); }
          }
      }
      if (nc <= kBruteChains) {
        if (best_w < 0) best = 2;
        side_best.assign(nc, 0);
        for (int c = 0; c < nc; ++c) side_best[c] = (char)(best >> c & 1);
      } else if (best_w < 0) {   // nothing balanced was found: the largest chain against the rest
        side_best.assign(nc, 0);
        side_best[(int)(std::max_element(size.begin(), size.end()) - size.begin())] = 1;
      }
      if (side_best[0]) for (auto& b : side_best) b ^= 1;   // chain 0 on side 0, as the enumeration has it
      for (int v : vars) side[v] = (int)side_best[side[v]];
    } else if (chains.size() >= 2) {
      nparts = (int)chains.size();
      for (int v : vars) side[v] = (int)(std::lower_bound(chains.begin(), chains.end(), chain_of[v >> 1]) - chains.begin());
    } else {
      std::vector<int> pos;
      for (int v : vars) pos.push_back(v >> 1);
      std::sort(pos.begin(), pos.end());
      pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
      if (pos.size() < 2) { leaf(vars, parent); return; }
      const int m = pos[pos.size() / 2];
      for (int v : vars) side[v] = (v >> 1) >= m;
    }
    // ---- greedy vertex cover of the couplings that cross the cut: highest remaining crossing degree first
    //      (ties: pose blocks before speed-bias blocks, then the lower variable index) — deterministic
    ++regid;
    for (int v : vars) { reg[v] = regid; inS[v] = 0; }
    std::priority_queue<std::tuple<int, int, int>> heap;
    for (int v : vars) {
      int d = 0;
      for (int w : adj[v]) d += (reg[w] == regid && side[w] != side[v]) ? 1 : 0;
      deg[v] = d;
      if (d) heap.push(std::make_tuple(d, 1 - (v & 1), -v));
    }
    std::vector<int> S;
    while (!heap.empty()) {
      const auto top = heap.top(); heap.pop();
      const int v = -std::get<2>(top);
      if (inS[v] || std::get<0>(top) != deg[v] || deg[v] == 0) continue;
      inS[v] = 1; S.push_back(v);
      for (int w : adj[v])
        if (reg[w] == regid && side[w] != side[v] && !inS[w] && deg[w] > 0) { if (--deg[w]) heap.push(std::make_tuple(deg[w], 1 - (w & 1), -w)); }
      deg[v] = 0;
    }
    // prune: a separator vertex all of whose crossing neighbours are in the separator too is redundant (greedy covers
    // contain such vertices); latest picks first, one pass — deterministic
    for (size_t q = S.size(); q-- > 0;) {
      const int v = S[q];
      bool needed = false;
      for (int w : adj[v]) if (reg[w] == regid && side[w] != side[v] && !inS[w]) { needed = true; break; }
      if (!needed) { inS[v] = 0; S[q] = -1; }
    }
    S.erase(std::remove(S.begin(), S.end(), -1), S.end());
    std::vector<std::vector<int>> parts(nparts);
    for (int v : vars) if (!inS[v]) parts[side[v]].push_back(v);
    const int n = S.empty() ? parent : new_node(S, parent);  // parts that do not touch: no separator node
    for (auto& pt : parts) build(pt, n);
  }
};

}  // namespace

// the coupling graph over the variables: built once per problem, shared by every candidate plan
static void nd_graph(int K, bool vi, int nchains, const int* chain_ptr, int npairs, const int* pair_i, const int* pair_j, int nepairs, const int* epair_i,
                     const int* epair_j, std::vector<std::vector<int>>& adj, std::vector<int>& chain_of) {
  chain_of.assign(K, 0);
  for (int c = 0; c < nchains; ++c) for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) chain_of[q] = c;
  adj.assign(2 * (size_t)K, {});
  auto link = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
  for (int p = 0; p < npairs; ++p) link(2 * pair_i[p], 2 * pair_j[p]);
  for (int p = 0; p < nepairs; ++p) link(2 * epair_i[p], 2 * epair_j[p]);
  if (vi)
    for (int c = 0; c < nchains; ++c)
      for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) {
        link(2 * q + 1, 2 * q);
        if (q > chain_ptr[c]) {
          const int v[4] = {2 * (q - 1), 2 * (q - 1) + 1, 2 * q, 2 * q + 1};
          for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) link(v[a], v[b]);
        }
      }
  for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
}

static bool nd_plan_build_mode(int top_mode, double group_frac, int K, bool vi, const std::vector<std::vector<int>>& adj, const std::vector<int>& chain_of,
                               int leaf_dims, NdHostPlan& out) {
  out = NdHostPlan();
  out.K = K; out.vi = vi ? 1 : 0; out.nvar = 2 * K;
  out.vnode.assign(2 * (size_t)K, -1); out.voff.assign(2 * (size_t)K, 0); out.vord.assign(2 * (size_t)K, -1);
  if (K <= 0) return false;
  // ---- tree
  std::vector<int> all;
  for (int q = 0; q < K; ++q) { all.push_back(2 * q); if (vi) all.push_back(2 * q + 1); }
  Builder bld(K, vi, leaf_dims, adj, chain_of, out);
  bld.top_mode = top_mode; bld.group_frac = group_frac;
  bld.build(all, -1);
  const int nn = out.nnodes;
  auto is_proper_ancestor = [&](int a, int n) {  // a above n?
    if (out.depth[a] >= out.depth[n]) return false;
    while (out.depth[n] > out.depth[a]) n = out.parent[n];
    return n == a;
  };
  // ---- panel balance. The levels (node heights) run one after the other and a level costs one serial panel chain per 256
  //      columns of its WIDEST front. Where the widest fronts of a level exceed a panel boundary by little, the excess
  //      unknowns move up into the parent's front (an ancestor's separator may always take more vertices: they leave the
  //      child's region) — if that does not push the parent's level over a boundary of its own.
  {
    std::vector<int> hgt(nn, 0);
    int nlev = 0;
    for (int n = nn - 1; n >= 0; --n) { for (int c : out.child[n]) hgt[n] = std::max(hgt[n], hgt[c] + 1); nlev = std::max(nlev, hgt[n] + 1); }
    auto panels = [](int d) { return std::max(1, (d + 255) / 256); };
    std::vector<int> levc(nlev, 1);
    for (int n = 0; n < nn; ++n) levc[hgt[n]] = std::max(levc[hgt[n]], panels(out.own_dims[n]));
    for (int l = 0; l + 1 < nlev; ++l) {
      const int m = levc[l];
      if (m < 2) continue;
      std::vector<std::pair<int, int>> moves;  // (node, number of trailing own variables that move up)
      std::vector<int> extra(nn, 0);
      bool ok = true;
      for (int n = 0; n < nn && ok; ++n) {
        if (hgt[n] != l || panels(out.own_dims[n]) < m) continue;
        const int need = out.own_dims[n] - 256 * (m - 1);
        int cnt = 0, mv = 0;
        for (size_t q = out.own[n].size(); q-- > 0 && mv < need;) { mv += NdHostPlan::vdim(out.own[n][q]); ++cnt; }
        if (need > 64 || out.parent[n] < 0 || cnt >= (int)out.own[n].size()) { ok = false; break; }
        moves.emplace_back(n, cnt); extra[out.parent[n]] += mv;
      }
      for (int n = 0; n < nn && ok; ++n) if (extra[n] && panels(out.own_dims[n] + extra[n]) > levc[hgt[n]]) ok = false;
      if (!ok) continue;
      for (auto& mvp : moves) {
        const int n = mvp.first, pn = out.parent[n];
        std::vector<int> up(out.own[n].end() - mvp.second, out.own[n].end());
        out.own[n].resize(out.own[n].size() - mvp.second);
        for (int v : up) out.own[pn].push_back(v);
        for (int node : {n, pn}) {
          int off = 0, ord = 0;
          for (int v : out.own[node]) { out.vnode[v] = node; out.voff[v] = off; out.vord[v] = ord++; off += NdHostPlan::vdim(v); }
          out.own_dims[node] = off;
        }
      }
      levc[l] = m - 1;
    }
  }
  // ---- symbolic factorisation, children before parents (a child's index is always above its parent's)
  out.strct.assign(nn, {});
  out.st_dims.assign(nn, 0);
  std::vector<int> stamp(2 * (size_t)K, -1);
  for (int n = nn - 1; n >= 0; --n) {
    std::vector<int>& st = out.strct[n];
    auto add = [&](int w) { if (stamp[w] != n) { stamp[w] = n; st.push_back(w); } };
    for (int v : out.own[n])
      for (int w : adj[v]) {
        const int m = out.vnode[w];
        if (m == n) continue;
        if (is_proper_ancestor(m, n)) add(w);
        else if (!is_proper_ancestor(n, m)) return false;  // separator property violated: a coupling joins two branches
      }
    for (int c : out.child[n])
      for (int w : out.strct[c]) if (out.vnode[w] != n) add(w);
    std::sort(st.begin(), st.end(), [&](int a, int b) {  // nearest ancestor first, then its own order
      return std::make_pair(-out.depth[out.vnode[a]], out.vord[a]) < std::make_pair(-out.depth[out.vnode[b]], out.vord[b]);
    });
    int d = 0;
    for (int w : st) d += NdHostPlan::vdim(w);
    out.st_dims[n] = d;
  }
  // ---- batches: nodes of equal height
  out.level.assign(nn, 0); out.slot.assign(nn, 0);
  for (int n = nn - 1; n >= 0; --n)
    for (int c : out.child[n]) out.level[n] = std::max(out.level[n], out.level[c] + 1);
  out.nlev = 0; out.maxdepth = 0;
  for (int n = 0; n < nn; ++n) { out.nlev = std::max(out.nlev, out.level[n] + 1); out.maxdepth = std::max(out.maxdepth, out.depth[n] + 1); }
  out.lev_nodes.assign(out.nlev, {});
  for (int n = 0; n < nn; ++n) { out.slot[n] = (int)out.lev_nodes[out.level[n]].size(); out.lev_nodes[out.level[n]].push_back(n); }
  out.lev_nI.assign(out.nlev, 0); out.lev_nO.assign(out.nlev, 0); out.lev_ntot.assign(out.nlev, 0);
  for (int l = 0; l < out.nlev; ++l) {
    int mo = 0, ms = 0;
    for (int n : out.lev_nodes[l]) { mo = std::max(mo, out.own_dims[n]); ms = std::max(ms, out.st_dims[n]); }
    out.lev_nI[l] = std::max(256, ((mo + 255) / 256) * 256);
    out.lev_nO[l] = ((ms + 127) / 128) * 128;
    out.lev_ntot[l] = out.lev_nI[l] + out.lev_nO[l];
    out.front_elems += (size_t)out.lev_nodes[l].size() * out.lev_ntot[l] * out.lev_ntot[l];
  }
  for (int n = 0; n < nn; ++n) {
    const double m = out.own_dims[n], b = out.st_dims[n];
    out.flops += m * m * m / 3.0 + m * m * b + m * b * b;
  }
  return true;
}

// What one factorisation of the plan costs on the device, roughly (round 5 figures of one MI355X): the levels run one after the other, a level
// costs one serial chain of ~140 us per 256 columns of its widest front plus ~40 us of transition, and the flops run at ~30 TFLOP/s beside them.
static double nd_plan_cost(const NdHostPlan& hp) {
  double t = hp.flops / 30e12;
  for (int l = 0; l < hp.nlev; ++l) t += (hp.lev_nI[l] / 256) * 140e-6 + 40e-6;
  return t;
}

bool nd_plan_build(int K, bool vi, int nchains, const int* chain_ptr, int npairs, const int* pair_i, const int* pair_j, int nepairs,
                   const int* epair_i, const int* epair_j, int leaf_dims, NdHostPlan& out, int top_mode) {
  // Candidates: (how a region of three or more agents is cut) x (the order below which a region is not cut further). All are built — the coupling
  // graph once, the trees in host threads, milliseconds each — and the cheapest by nd_plan_cost is kept (ties: the first in the fixed order below).
  //   top: ONE cover of all cross-agent couplings (0) | two groups of agents, recursively (1). 5-agent map: 2.84e10 flops / 23 serial panels /
  //        7 levels against 2.42e10 / 22 / 9; 12-agent map: a 17 652-order root and 2.13e12 flops against a 7 206-order root and 1.30e12.
  //   leaf: where the widest fronts land relative to the 256-column panel boundaries decides the number of serial panels. Measured on one
  //        MI355X (profiles/r05_ab_experiments.txt), factor+solve per iteration against the model: 5-agent map leaf 400: 3.56 ms (model 3.96),
  //        900: 3.77 (4.10), 500: 3.73 (4.11), 600: 3.90 (4.25), 1200: 4.04 (4.29), 250-350: 3.8-4.0 (4.28-4.41); mh01 and the 3-agent map are
  //        fastest at 600, as the model says (1.39 / 2.24 ms against 1.57 / 2.42 at 400). Rounds 3-4 fixed 600.
  // top_mode 0 (the plan of a SHARDED solve): one separator of all agents at the top, whose children (one region per agent) are the subtrees
  // dealt to the ranks. COVGPU_ND_TOP=0 / 1 and leaf_dims > 0 (COVGPU_ND_LEAF) force a choice. Deterministic: every rank of a sharded solve
  // and covgpu_shard_plan pick the same plan.
  std::vector<std::vector<int>> adj;
  std::vector<int> chain_of;
  out = NdHostPlan();
  out.K = K; out.vi = vi ? 1 : 0; out.nvar = 2 * K;
  if (K <= 0) return false;
  nd_graph(K, vi, nchains, chain_ptr, npairs, pair_i, pair_j, nepairs, epair_i, epair_j, adj, chain_of);
  int forced = top_mode;
  if (const char* e = getenv("COVGPU_ND_TOP")) forced = atoi(e) != 0 ? 1 : 0;
  std::vector<int> modes, leaves;
  if (nchains < 3 || forced == 0 || leaf_dims >= (1 << 29)) modes = {0};
  else if (forced == 1) modes = {1};
  else modes = {0, 1};
  if (leaf_dims > 0) leaves = {leaf_dims};
  else leaves = {600, 400, 500, 900};
  // two groups: how unequal they may be. A third of the unknowns on the lighter side finds the lightest cut; 40 % costs a heavier root and buys balance —
  // on the 12-agent map a 7 788- instead of a 7 206-order root, 69 instead of 80 serial panels and 1.18e12 instead of 1.30e12 flops (the fronts with
  // 7 000-unknown borders sit on the heavier side), and for a sharded solve 61 % instead of 81 % of the flops on the busier of two ranks.
  std::vector<double> fracs = {3.0, 2.5};
  if (const char* e = getenv("COVGPU_ND_GROUP_FRAC")) fracs = {std::max(2.0, atof(e))};
  struct Cand { int mode, leaf; double frac; NdHostPlan hp; bool ok = false; double cost = 0; };
  std::vector<Cand> cand;
  for (int lf : leaves)
    for (int m : modes)
      for (double fr : (m == 1 ? fracs : std::vector<double>{3.0})) { cand.emplace_back(); cand.back().mode = m; cand.back().leaf = lf; cand.back().frac = fr; }
  // (a worker that throws — std::bad_alloc on a huge map — marks its candidate failed instead of reaching std::terminate; threads that cannot be
  //  created leave their candidates to the calling thread)
  auto run = [&](size_t i) {
    try {
      cand[i].ok = nd_plan_build_mode(cand[i].mode, cand[i].frac, K, vi, adj, chain_of, cand[i].leaf, cand[i].hp);
      if (cand[i].ok) cand[i].cost = nd_plan_cost(cand[i].hp);
    } catch (...) { cand[i].ok = false; }
  };
  if (cand.size() == 1) run(0);
  else {
    std::vector<std::thread> th;
    std::vector<char> started(cand.size(), 0);
    for (size_t i = 1; i < cand.size(); ++i) {
      try { th.emplace_back(run, i); started[i] = 1; } catch (...) { break; }
    }
    run(0);
    for (auto& t : th) t.join();
    for (size_t i = 1; i < cand.size(); ++i) if (!started[i]) run(i);
  }
  int best = -1;
  for (size_t i = 0; i < cand.size(); ++i) if (cand[i].ok && (best < 0 || cand[i].cost < cand[best].cost)) best = (int)i;
  if (best < 0) return false;
  out = std::move(cand[best].hp);
  out.top_mode = cand[best].mode; out.leaf = cand[best].leaf; out.group_frac100 = (int)(cand[best].frac * 100.0 + 0.5);
  return true;
}

double nd_shard_cost(const NdHostPlan& hp, int world) {
  const int nn = hp.nnodes;
  double top_fl = 0.0, exch = 0.0;
  std::vector<double> rank_fl(std::max(world, 1), 0.0);
  for (int n = 0; n < nn; ++n) {
    const double m = hp.own_dims[n], b = hp.st_dims[n], fl = m * m * m / 3.0 + m * m * b + m * b * b;
    if (hp.node_rank.empty() || hp.node_rank[n] < 0) { top_fl += fl; exch += 0.5 * (m + b) * (m + b) * 8.0; }
    else rank_fl[hp.node_rank[n]] += fl;
  }
  double busiest = 0.0;
  for (double f : rank_fl) busiest = std::max(busiest, f);
  double chain = 0.0;
  for (int l = 0; l < hp.nlev; ++l) chain += (hp.lev_nI[l] / 256) * 140e-6 + 40e-6;
  return (top_fl + busiest) / 30e12 + chain + (world > 1 ? 2.0 * (world - 1) / world * exch / 150e9 : 0.0);
}

void nd_shard_assign(NdHostPlan& hp, int world, double top_cap_bytes) {
  const int nn = hp.nnodes;
  hp.node_rank.assign(nn, -1);
  hp.nsub = 0;
  std::vector<double> w(nn, 0.0), sub(nn, 0.0);
  for (int n = 0; n < nn; ++n) { const double m = hp.own_dims[n], b = hp.st_dims[n]; w[n] = m * m * m / 3.0 + m * m * b + m * b * b + 1.0; }
  for (int n = nn - 1; n >= 0; --n) { sub[n] += w[n]; if (hp.parent[n] >= 0) sub[hp.parent[n]] += sub[n]; }  // children have higher indices
  std::vector<char> top(nn, 0);
  std::vector<int> cand;
  double top_bytes = 0.0;
  for (int n = 0; n < nn; ++n)
    if (hp.parent[n] < 0) { top[n] = 1; top_bytes += 8.0 * (double)hp.own_dims[n] * hp.own_dims[n]; for (int c : hp.child[n]) cand.push_back(c); }
  for (int it = 0; it < 256; ++it) {
    if (cand.empty()) break;
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return sub[a] != sub[b] ? sub[a] > sub[b] : a < b; });
    double tot = 0.0;
    for (int c : cand) tot += sub[c];
    // enough pieces and none much heavier than a rank's fair share: stop. Opening a node makes it a TOP node — its whole front
    // joins the all-reduced range and its factorisation is replicated on every rank. The solve is latency-bound (a rank runs
    // two agents' subtrees in the same batched launches almost as fast as one), so what a larger top buys in balance it loses in
    // the exchange: the top never grows beyond 48 MiB of fronts unless the roots alone are larger — rather fewer subtrees than
    // ranks (ranks without a subtree still hold the top and take part in the exchange).
    if ((int)cand.size() >= world && sub[cand[0]] <= 1.25 * tot / world) break;
    int pick = -1;
    for (size_t q = 0; q < cand.size() && pick < 0; ++q) {   // heaviest that can still be opened within the cap
      const int c = cand[q];
      const double nb = 8.0 * (double)(hp.own_dims[c] + hp.st_dims[c]) * (double)(hp.own_dims[c] + hp.st_dims[c]);
      if (!hp.child[c].empty() && top_bytes + nb <= top_cap_bytes) pick = (int)q;
    }
    if (pick < 0) break;
    const int c = cand[pick];
    cand.erase(cand.begin() + pick);
    top[c] = 1; top_bytes += 8.0 * (double)(hp.own_dims[c] + hp.st_dims[c]) * (double)(hp.own_dims[c] + hp.st_dims[c]);
    for (int d : hp.child[c]) cand.push_back(d);
  }
  std::sort(cand.begin(), cand.end(), [&](int a, int b) { return sub[a] != sub[b] ? sub[a] > sub[b] : a < b; });
  std::vector<double> load(world, 0.0);
  std::vector<int> owner(nn, -1);
  for (int c : cand) {
    int best = 0;
    for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
    owner[c] = best; load[best] += sub[c];
  }
  hp.nsub = (int)cand.size();
  for (int n = 0; n < nn; ++n) {  // parents come first: a node below a dealt subtree root inherits its rank
    if (top[n]) { hp.node_rank[n] = -1; continue; }
    hp.node_rank[n] = owner[n] >= 0 ? owner[n] : hp.node_rank[hp.parent[n]];
  }
}

void nd_plan_remap(NdHostPlan& hp, const std::vector<int>& var_map) {
  std::vector<int> vnode(hp.vnode.size(), -1), voff(hp.voff.size(), 0), vord(hp.vord.size(), -1);
  for (size_t v = 0; v < hp.vnode.size(); ++v)
    if (hp.vnode[v] >= 0) { const int u = var_map[v]; vnode[u] = hp.vnode[v]; voff[u] = hp.voff[v]; vord[u] = hp.vord[v]; }
  hp.vnode.swap(vnode); hp.voff.swap(voff); hp.vord.swap(vord);
  for (auto& l : hp.own) for (int& v : l) v = var_map[v];
  for (auto& l : hp.strct) for (int& v : l) v = var_map[v];
}

}  // namespace covgpu
