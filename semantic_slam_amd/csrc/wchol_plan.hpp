// Host-side symbolic phase of the WINDOW multifrontal block Cholesky (round 3; no HIP types: also used by the CPU-side tests).
//
// Replaces the symbolic half of g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with "lm_var"
// (reference src/ps_graph_slam/graph_slam.cpp:27,67-73).  Numeric kernels: sslam_wchol.hip.
//
// Shape (what changed against the piece plan of chol_plan.hpp, which stays for the multi right-hand-side solves of the marginals):
//   * the elimination tree (minimum-degree order on the block graph) is cut into SEGMENTS -- connected parts of the tree whose
//     columns one workgroup eliminates one after the other, children before parents;
//   * the workgroup keeps the ACTIVE SUBMATRIX of its segment on chip: a *window* of at most W block rows (the rows some
//     eliminated column has touched and that are not eliminated yet), every 6 x 6 tile of the window's lower triangle owned by four
//     lanes (3 x 3 quarters) IN REGISTERS.  A row takes the lowest free slot when it is first touched and gives it back when it is
//     eliminated, so a chain of the tree slides through the window: the update matrix of a column never leaves the registers while
//     its parent is in the same segment (the multifrontal hand-over through HBM of the piece plan only remains at segment borders);
//   * per pivot column the only index data is one 32-byte step record + one 16-byte record per row of the column: no per-update
//     records at all -- the tile (a, b) of the window is updated by  L(a, c) L(b, c)^T  for every pair of rows a, b of the pivot
//     column c, addressed by slot numbers;
//   * the factor leaves as one contiguous panel per column (diagonal block first, then the rows of the column), each written once;
//     the right-hand side rides along as an extra row of the window (forward substitution fused);
//   * segments of equal depth in the segment tree share a launch; two classes of segments (window of <= 10 rows: one wave;
//     larger windows: four waves) so that the many small fronts at the bottom of the tree do not pay for the few big ones at its top.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "chol_plan.hpp"   // SymIn / SymGraph, chol_detail::min_degree

namespace sslam {

struct WStep {       // one pivot column
  int piv;           // pivot slot | dim << 8 | rows of the column (pivot included) << 16
  int row0;          // first WRow record (the pivot's own record first)
  int loff;          // Lval offset (doubles) of the column panel
  int xoff;          // offset of the column's unknowns in the internal row order (b, y, x)
  int child0, nchild;   // update matrices of child segments that join here (WChild records)
  unsigned mask_lo, mask_hi;   // window slots of the column's rows (pivot included), 64 bits
};
struct WRow {        // one row of a pivot column
  int slot;          // window slot | dim << 8 | (H block stored transposed) << 16
  int hsrc;          // offset of the H block (row, column) in the [H || b] buffer, -1: fill-in
  int lofs;          // offset of the row's block inside the column panel (doubles)
  int xoff;          // offset of the row's unknowns
};
struct WChild { int uoff, m, map0, pad; };   // update matrix of a child segment: Uval offset, rows, first byte of its slot map (wmap)
struct WFin { int slot, xoff; };            // a row still in the window when the segment ends (slot | dim << 8), ascending slot
struct WSeg {
  int graph, step0, nsteps;
  int uoff, m, fin0;          // update matrix handed to the parent segment (-1: root), its rows, first WFin record
  int cls, level;
};
// update matrix of a segment with m rows (ascending slot order = local index p): 6 x 6 tile (p, q), p >= q, rows of p x rows of q,
// padded, at uoff + 36 * (p (p + 1) / 2 + q); right-hand-side part of row p at uoff + 36 * m (m + 1) / 2 + 6 p
inline int64_t wchol_u_doubles(int m) { return 36 * ((int64_t)m * (m + 1) / 2) + 6 * (int64_t)m; }

struct WClass { int nt, S, wmax; };        // threads, tile registers per thread (tiles per thread / 4 lanes per tile), window slots
// tiles of a window of w slots: w (w + 1) / 2 <= S * nt / 4 ;  rows of a column need 6 * rows + 1 threads in the solve phase
constexpr WClass kWClass[3] = {{64, 4, 10}, {512, 6, 38}, {1024, 6, 54}};

struct WOpts {
  int colcap[3] = {48, 192, 4096};   // columns per segment, by class
  int w_small = kWClass[0].wmax;
  bool dump = false;
  void from_env(int B) {
    auto ei = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    if (B < 32) { colcap[0] = 12; colcap[1] = 64; }   // few graphs: more, shorter segments (latency of the segment chain)
    colcap[0] = std::max(1, ei("SSLAM_WCHOL_COLCAP0", colcap[0]));
    colcap[1] = std::max(1, ei("SSLAM_WCHOL_COLCAP1", colcap[1]));
    w_small = std::min(kWClass[0].wmax, std::max(2, ei("SSLAM_WCHOL_WSMALL", w_small)));
    dump = getenv("SSLAM_CHOL_DUMP") != nullptr;
  }
};

struct WHost {
  int B = 0, dim = 0, ncol = 0, nlevels = 0;
  int64_t lnz = 0, unz = 0;
  std::vector<WStep> step; std::vector<WRow> row; std::vector<WChild> child; std::vector<WFin> fin; std::vector<WSeg> seg;
  std::vector<unsigned char> wmap;
  std::vector<int> launch_ptr, launch_cls;   // segments are stored in launch order: launch l = segments [launch_ptr[l], launch_ptr[l + 1]) of class launch_cls[l]
  int max_rows[3] = {0, 0, 0}, max_children[3] = {0, 0, 0}, max_window[3] = {0, 0, 0};
  std::string error;
};

namespace wchol_detail {

struct GraphPlan {   // everything of one graph, local offsets (made global when the graphs are concatenated)
  std::vector<WStep> step; std::vector<WRow> row; std::vector<WChild> child; std::vector<WFin> fin; std::vector<WSeg> seg;
  std::vector<unsigned char> wmap;
  int64_t lnz = 0, unz = 0;
  int nlevels = 0;
  int max_rows[3] = {0, 0, 0}, max_children[3] = {0, 0, 0}, max_window[3] = {0, 0, 0};
  std::string error;
};

// window needed to run the column sequence `seq` in one workgroup: max over the steps of |rows touched so far and not eliminated|
inline int window_max(const std::vector<int>& seq, const std::vector<std::vector<int>>& cs, std::vector<int>& stamp, int& tick) {
  ++tick;
  int live = 0, wmax = 0;
  for (int j : seq) {
    if (stamp[j] != tick) { stamp[j] = tick; ++live; }
    for (int r : cs[j]) if (stamp[r] != tick) { stamp[r] = tick; ++live; }
    wmax = std::max(wmax, live);
    --live;   // j is eliminated
  }
  return wmax;
}

inline void plan_graph(const SymIn& in, int g, const WOpts& opt, const std::unordered_map<uint64_t, int>& hoff, GraphPlan& out) {
  using chol_detail::GraphSym;
  const int nPr = in.nPr;
  const SymGraph& sg = in.seg[g];
  const int n = sg.nprow + sg.nlrow;
  auto loc2row = [&](int v) { return v < sg.nprow ? sg.prow0 + v : nPr + sg.lrow0 + (v - sg.nprow); };
  auto row_dim = [&](int r) { return r < nPr ? 6 : 3; };
  auto row_xoff = [&](int r) { return r < nPr ? 6 * r : 6 * nPr + 3 * (r - nPr); };
  auto key = [](int a, int c) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)c; };
  // ---- adjacency of this graph (local ids), minimum-degree order, column structures by elimination position
  std::vector<std::vector<int>> ladj(n);
  {
    auto row2loc = [&](int r) { return r < nPr ? r - sg.prow0 : sg.nprow + (r - nPr - sg.lrow0); };
    for (int k = sg.pp0; k < sg.pp1; ++k) { const int a = row2loc(in.ppoff[k].first), c = row2loc(in.ppoff[k].second); ladj[a].push_back(c); ladj[c].push_back(a); }
    for (int k = sg.pl0; k < sg.pl1; ++k) { const int a = row2loc(in.plblk[k].first), c = row2loc(nPr + in.plblk[k].second); ladj[a].push_back(c); ladj[c].push_back(a); }
    for (int k = sg.ll0; k < sg.ll1; ++k) { const int a = row2loc(nPr + in.llblk[k].first), c = row2loc(nPr + in.llblk[k].second); ladj[a].push_back(c); ladj[c].push_back(a); }
  }
  GraphSym S;
  chol_detail::min_degree(n, ladj, S);
  std::vector<int> pos(n);
  for (int s = 0; s < n; ++s) pos[S.order[s]] = s;
  std::vector<std::vector<int>> cs(n);      // rows of column s (positions > s), ascending
  std::vector<int> parent(n, -1), grow(n), dimc(n);
  for (int s = 0; s < n; ++s) {
    const int v = S.order[s];
    grow[s] = loc2row(v); dimc[s] = row_dim(grow[s]);
    for (int q = S.cs_start[v]; q < S.cs_start[v] + S.cs_len[v]; ++q) cs[s].push_back(pos[S.cs_idx[q]]);
    std::sort(cs[s].begin(), cs[s].end());
    if (!cs[s].empty()) {
      if (cs[s][0] <= s) { out.error = "symbolic factorisation inconsistent (row not below its column)"; return; }
      parent[s] = cs[s][0];
    }
  }
  std::vector<std::vector<int>> kids(n);
  for (int s = 0; s < n; ++s) if (parent[s] >= 0) kids[parent[s]].push_back(s);
  // ---- segments: bottom-up; a column extends the open segments of its children while the window and column caps hold
  struct SegB { std::vector<int> cols; int cls = 0; bool open = true; };
  std::vector<SegB> segs;
  std::vector<int> seg_of(n, -1), stamp(n, 0);
  int tick = 0;
  for (int s = 0; s < n; ++s) {
    const int need = 1 + (int)cs[s].size();
    int mincls = need <= opt.w_small ? 0 : (need <= kWClass[1].wmax ? 1 : 2);
    if (need > kWClass[2].wmax) { out.error = "a column of the factor has more than " + std::to_string(kWClass[2].wmax - 1) + " off-diagonal blocks"; return; }
    std::vector<int> cand;
    for (int c : kids[s]) { const int q = seg_of[c]; if (segs[q].open && segs[q].cols.back() == c) cand.push_back(q); }
    // the class of the merged segment: the highest among the column's own need and the children that are taken
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
      if (segs[a].cls != segs[b].cls) return segs[a].cls > segs[b].cls;
      return segs[a].cols.size() != segs[b].cols.size() ? segs[a].cols.size() > segs[b].cols.size() : a < b;
    });
    std::vector<int> take;
    int cls = mincls;
    {
      // greedy: add candidates one by one (largest first) while the merged sequence still fits the class
      std::vector<int> seq;
      for (int q : cand) {
        const int c2 = std::max(cls, segs[q].cls);
        // a small segment is not pulled into a bigger class unless the column itself already needs that class
        if (segs[q].cls < mincls && segs[q].cols.size() > 4) continue;
        std::vector<int> trial;
        for (int t : take) trial.insert(trial.end(), segs[t].cols.begin(), segs[t].cols.end());
        trial.insert(trial.end(), segs[q].cols.begin(), segs[q].cols.end());
        trial.push_back(s);
        const int wlim = c2 == 0 ? opt.w_small : kWClass[c2].wmax;
        if ((int)trial.size() > opt.colcap[c2]) continue;
        if (window_max(trial, cs, stamp, tick) > wlim) continue;
        take.push_back(q); cls = c2;
      }
    }
    for (int q : cand) segs[q].open = false;
    int id;
    if (take.empty()) {
      id = (int)segs.size();
      segs.push_back(SegB{{s}, cls, true});
    } else {
      id = take[0];
      std::vector<int> seq;
      for (int t : take) seq.insert(seq.end(), segs[t].cols.begin(), segs[t].cols.end());
      seq.push_back(s);
      for (size_t k = 1; k < take.size(); ++k) { segs[take[k]].cols.clear(); segs[take[k]].open = false; }
      segs[id].cols = std::move(seq); segs[id].cls = cls; segs[id].open = true;
      for (int c : segs[id].cols) seg_of[c] = id;
    }
    seg_of[s] = id;
  }
  // ---- live segments in order of their root (children before parents); external children per column
  std::vector<int> ids;
  for (int q = 0; q < (int)segs.size(); ++q) if (!segs[q].cols.empty()) ids.push_back(q);
  std::sort(ids.begin(), ids.end(), [&](int a, int b) { return segs[a].cols.back() < segs[b].cols.back(); });
  std::vector<int> rank(segs.size(), -1);
  for (size_t k = 0; k < ids.size(); ++k) rank[ids[k]] = (int)k;
  std::vector<std::vector<int>> ext(n);   // column -> child segments (rank) whose root's parent it is
  for (int q : ids) { const int root = segs[q].cols.back(); if (parent[root] >= 0) ext[parent[root]].push_back(rank[q]); }
  const int nseg = (int)ids.size();
  std::vector<std::vector<int>> fin_rows(nseg);   // per segment (rank): the rows of its update matrix, ascending slot
  std::vector<int> level(nseg, 0);
  out.seg.assign(nseg, WSeg{});
  std::vector<int> slot_of(n, -1);
  for (int k = 0; k < nseg; ++k) {
    const SegB& sb = segs[ids[k]];
    WSeg& ws = out.seg[k];
    ws.graph = g; ws.step0 = (int)out.step.size(); ws.nsteps = (int)sb.cols.size(); ws.cls = sb.cls; ws.level = 0;
    const int W = kWClass[sb.cls].wmax;
    std::vector<int> slot_row(W, -1);
    std::vector<int> touched;
    int live = 0;
    auto take_slot = [&](int r) {
      for (int q = 0; q < W; ++q) if (slot_row[q] < 0) { slot_row[q] = r; slot_of[r] = q; touched.push_back(r); ++live; return q; }
      return -1;
    };
    for (int j : sb.cols) {
      WStep st{};
      st.row0 = (int)out.row.size();
      st.loff = (int)out.lnz; st.xoff = row_xoff(grow[j]);
      st.child0 = (int)out.child.size();
      const int dj = dimc[j];
      int lofs = 0;
      unsigned long long mask = 0;
      std::vector<int> rows{j};
      rows.insert(rows.end(), cs[j].begin(), cs[j].end());
      for (int r : rows) {
        int q = slot_of[r];
        if (q < 0) q = take_slot(r);
        if (q < 0) { out.error = "window overflow (segment cut inconsistent)"; return; }
        mask |= 1ull << q;
        WRow wr{};
        const int di = dimc[r];
        int hsrc, fmt = 0;
        if (r == j) hsrc = grow[j] < nPr ? grow[j] * 36 : (int)(in.hll_base + (int64_t)(grow[j] - nPr) * 9);
        else {
          const int ri = grow[r], rj = grow[j];
          const int a = std::min(ri, rj), c = std::max(ri, rj);
          auto it = hoff.find(key(a, c));
          if (it == hoff.end()) hsrc = -1; else { hsrc = it->second; fmt = ri == a ? 0 : 1; }   // stored [min][max]; needed [row][column]
        }
        wr.slot = q | (di << 8) | (fmt << 16); wr.hsrc = hsrc; wr.lofs = lofs; wr.xoff = row_xoff(grow[r]);
        lofs += di * dj;
        out.row.push_back(wr);
      }
      st.piv = slot_of[j] | (dj << 8) | ((int)rows.size() << 16);
      st.mask_lo = (unsigned)(mask & 0xFFFFFFFFull); st.mask_hi = (unsigned)(mask >> 32);
      out.lnz += (lofs + 1) & ~1;   // panels start on 16-byte boundaries (the kernels store them with 16-byte stores)
      // update matrices of child segments that hang below this column
      for (int ck : ext[j]) {
        if (ck >= k) { out.error = "segment order inconsistent"; return; }
        WChild wc{};
        wc.uoff = out.seg[ck].uoff; wc.m = out.seg[ck].m; wc.map0 = (int)out.wmap.size();
        for (int r : fin_rows[ck]) {
          if (slot_of[r] < 0) { out.error = "a child's boundary row is not in the parent's window"; return; }
          out.wmap.push_back((unsigned char)slot_of[r]);
        }
        while (out.wmap.size() & 3) out.wmap.push_back(0xFF);
        out.child.push_back(wc);
        level[k] = std::max(level[k], level[ck] + 1);
      }
      st.nchild = (int)out.child.size() - st.child0;
      out.step.push_back(st);
      out.max_rows[sb.cls] = std::max(out.max_rows[sb.cls], (int)rows.size());
      out.max_children[sb.cls] = std::max(out.max_children[sb.cls], st.nchild);
      out.max_window[sb.cls] = std::max(out.max_window[sb.cls], live);
      // the pivot leaves the window
      slot_row[slot_of[j]] = -1; slot_of[j] = -1; --live;
    }
    // what is left is the update matrix for the parent segment
    ws.fin0 = (int)out.fin.size();
    for (int q = 0; q < W; ++q)
      if (slot_row[q] >= 0) { const int r = slot_row[q]; fin_rows[k].push_back(r); out.fin.push_back(WFin{q | (dimc[r] << 8), row_xoff(grow[r])}); }
    ws.m = (int)fin_rows[k].size();
    if (ws.m > 0) { ws.uoff = (int)out.unz; out.unz += wchol_u_doubles(ws.m); } else ws.uoff = -1;
    if (ws.m > 0 && parent[sb.cols.back()] < 0) { out.error = "a root segment has an update matrix"; return; }
    ws.level = level[k];
    out.nlevels = std::max(out.nlevels, level[k] + 1);
    for (int r : touched) slot_of[r] = -1;
  }
}

}  // namespace wchol_detail

// Plan of a whole batch.  Returns 0, or -1 with out.error set.
inline int wchol_symbolic(const SymIn& in, const WOpts& opt, WHost& out) {
  using namespace wchol_detail;
  out = WHost();
  const int B = in.B;
  out.B = B; out.dim = 6 * in.nPr + 3 * in.nLr;
  auto key = [](int a, int c) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)c; };
  std::unordered_map<uint64_t, int> hoff;
  hoff.reserve(in.ppoff.size() + in.plblk.size());
  for (size_t i = 0; i < in.ppoff.size(); ++i) hoff[key(in.ppoff[i].first, in.ppoff[i].second)] = (int)(in.hpp_off_base + (int64_t)i * 36);
  for (size_t i = 0; i < in.plblk.size(); ++i) hoff[key(in.plblk[i].first, in.nPr + in.plblk[i].second)] = (int)(in.hpl_base + (int64_t)i * 18);
  for (size_t i = 0; i < in.llblk.size(); ++i) hoff[key(in.nPr + in.llblk[i].first, in.nPr + in.llblk[i].second)] = (int)(in.hll_off_base + (int64_t)i * 9);
  // the off-diagonal blocks of every graph are a contiguous range of ppoff / plblk (batch_build emits them graph by graph)
  SymIn in2 = in;
  {
    size_t pp = 0, pl = 0, ll = 0;
    for (int g = 0; g < B; ++g) {
      SymGraph& sg = in2.seg[g];
      sg.pp0 = (int)pp;
      while (pp < in.ppoff.size() && in.ppoff[pp].first >= sg.prow0 && in.ppoff[pp].first < sg.prow0 + sg.nprow) ++pp;
      sg.pp1 = (int)pp;
      sg.pl0 = (int)pl;
      while (pl < in.plblk.size() && in.plblk[pl].first >= sg.prow0 && in.plblk[pl].first < sg.prow0 + sg.nprow) ++pl;
      sg.pl1 = (int)pl;
      sg.ll0 = (int)ll;
      while (ll < in.llblk.size() && in.llblk[ll].first >= sg.lrow0 && in.llblk[ll].first < sg.lrow0 + sg.nlrow) ++ll;
      sg.ll1 = (int)ll;
    }
    if (pp != in.ppoff.size() || pl != in.plblk.size() || ll != in.llblk.size()) { out.error = "off-diagonal blocks are not grouped by graph"; return -1; }
  }
  std::vector<GraphPlan> gp(B);
  {
    unsigned nth = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    if (const char* e = getenv("SSLAM_PLAN_THREADS")) nth = std::max(1, atoi(e));
    nth = std::min<unsigned>(nth, (unsigned)std::max(1, B));
    auto work = [&](unsigned t) { for (int g = (int)t; g < B; g += (int)nth) plan_graph(in2, g, opt, hoff, gp[g]); };
    if (nth <= 1) work(0);
    else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nth; ++t) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
  }
  // ---- concatenate, segments in launch order (level, class, graph)
  int nlev = 0;
  for (int g = 0; g < B; ++g) { if (!gp[g].error.empty()) { out.error = gp[g].error; return -1; } nlev = std::max(nlev, gp[g].nlevels); }
  out.nlevels = nlev;
  std::vector<int64_t> sbase(B), rbase(B), cbase(B), fbase(B), mbase(B), lbase(B), ubase(B);
  {
    int64_t s = 0, r = 0, c = 0, f = 0, m = 0, l = 0, u = 0;
    for (int g = 0; g < B; ++g) {
      sbase[g] = s; rbase[g] = r; cbase[g] = c; fbase[g] = f; mbase[g] = m; lbase[g] = l; ubase[g] = u;
      s += (int64_t)gp[g].step.size(); r += (int64_t)gp[g].row.size(); c += (int64_t)gp[g].child.size(); f += (int64_t)gp[g].fin.size();
      m += (int64_t)gp[g].wmap.size(); l += gp[g].lnz; u += gp[g].unz;
      for (int k = 0; k < 3; ++k) {
        out.max_rows[k] = std::max(out.max_rows[k], gp[g].max_rows[k]); out.max_children[k] = std::max(out.max_children[k], gp[g].max_children[k]);
        out.max_window[k] = std::max(out.max_window[k], gp[g].max_window[k]);
      }
    }
    if (l >= ((int64_t)1 << 31) - 4096 || u >= ((int64_t)1 << 31) - 4096 || r >= ((int64_t)1 << 31) - 4096) { out.error = "factor too large for int32 offsets"; return -1; }
    out.lnz = l; out.unz = u; out.ncol = (int)s;
    out.step.reserve(s); out.row.reserve(r); out.child.reserve(c); out.fin.reserve(f); out.wmap.reserve(m);
  }
  for (int g = 0; g < B; ++g) {
    const GraphPlan& P = gp[g];
    for (WStep st : P.step) { st.row0 += (int)rbase[g]; st.loff += (int)lbase[g]; st.child0 += (int)cbase[g]; out.step.push_back(st); }
    for (const WRow& r : P.row) out.row.push_back(r);
    for (WChild c : P.child) { c.uoff += (int)ubase[g]; c.map0 += (int)mbase[g]; out.child.push_back(c); }
    for (const WFin& f : P.fin) out.fin.push_back(f);
    out.wmap.insert(out.wmap.end(), P.wmap.begin(), P.wmap.end());
  }
  out.launch_ptr.assign(1, 0);
  for (int lv = 0; lv < nlev; ++lv)
    for (int cls = 0; cls < 3; ++cls) {
      const size_t before = out.seg.size();
      for (int g = 0; g < B; ++g)
        for (WSeg s : gp[g].seg)
          if (s.level == lv && s.cls == cls) {
            s.step0 += (int)sbase[g]; s.fin0 += (int)fbase[g];
            if (s.uoff >= 0) s.uoff += (int)ubase[g];
            out.seg.push_back(s);
          }
      if (out.seg.size() > before) { out.launch_ptr.push_back((int)out.seg.size()); out.launch_cls.push_back(cls); }
    }
  if (opt.dump) {
    fprintf(stderr, "[wchol-dump] B %d columns %d segments %zu lnz %lld unz %lld row records %zu levels %d launches %zu | max rows per column %d/%d/%d, window %d/%d/%d\n",
            B, out.ncol, out.seg.size(), (long long)out.lnz, (long long)out.unz, out.row.size(), nlev, out.launch_cls.size(),
            out.max_rows[0], out.max_rows[1], out.max_rows[2], out.max_window[0], out.max_window[1], out.max_window[2]);
    for (size_t l = 0; l < out.launch_cls.size(); ++l)
      fprintf(stderr, "[wchol-dump]   launch %zu: class %d, %d segments\n", l, out.launch_cls[l], out.launch_ptr[l + 1] - out.launch_ptr[l]);
  }
  return 0;
}

}  // namespace sslam
