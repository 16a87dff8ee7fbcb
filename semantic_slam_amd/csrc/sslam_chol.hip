// Sparse block Cholesky for the LM normal equations on gfx950 (FP64), numeric phase.
//
// Replaces g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with "lm_var"
// (reference src/ps_graph_slam/graph_slam.cpp:27,67-73).  The symbolic phase (ordering,
// elimination tree, pieces, work items) is host code in chol_plan.hpp; read its header for the design.  Here:
//   k_chol_pieces<NT>      one workgroup per piece of one depth of the piece tree: gather A + lambda I into LDS,
//                          external updates from HBM, internal levels out of LDS, one coalesced write of L and y
//   k_chol_tail<NT>        one workgroup per graph walks the top pieces of its tree in elimination order
//   k_chol_back_pieces / k_chol_back_tail   the same pieces top-down for x = L^-T y
//   k_chol_forward_level / k_chol_backward_level<Q>   level-scheduled multi right-hand-side solves (marginals)
// Everything is gather-form: every L entry is written by exactly one thread, sums run in a fixed order -> bitwise
// repeatable.  Forward substitution is fused into the factorisation (b rides along as an extra row of the diagonal block).
//   k_front_pieces / k_front_tail (front_kernels.hpp, round 6)   the same pieces through the front tables of front_plan.hpp: the default of batches >= 32
// Tuning: ONE environment variable, read when a plan is built -- SSLAM_CHOL_OPTS="key=value,..." with the field names of CholOpts
// (chol_plan.hpp).  The per-knob variables of rounds 2-4 (SSLAM_CHOL_CAP_LEAF, SSLAM_CHOL_FLOW, SSLAM_FLOW_DEFER, ...) are gone; setting one
// earns a warning on stderr, once, instead of a sweep that silently measures the default plan.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/sslam.h"
#include "graph_engine.hpp"
#include "chol_plan.hpp"

namespace sslam {

struct CholView {
  int ncol, nlevels, dim, npiece;
  const ColMeta* col;
  const BlkMeta* blk;
  const UpdMeta* upd;
  const ItemMeta* item;
  const MbMeta* mb;
  const ILevel* ilv;
  const PieceMeta* piece;
  const PieceMeta* lpiece;  // the same records in launch order: [pieces by depth | tails by graph]
  int ltail0;               // first tail record in lpiece
  const AsmSrc* asrc;       // child update-matrix blocks absorbed when a piece is gathered
  const AsmSrc* usrc;       // child update-matrix blocks passed on through a piece's own update matrix
  const UItem* uitem;
  const UMb* umb;
  const RCol* rcol;         // tail pieces, right-looking form: per column, its internal updates of later blocks of the piece ...
  const UpdMeta* rupd;      // ... (UpdMeta's right-looking form); nullptr: target-major items
  const FwdMeta* fwd;       // blocks of every row (multi right-hand-side forward substitution)
  const int* lvl_cols;      // columns grouped by level of the elimination tree
  const int* plv_pieces;    // pieces grouped by depth
  const int* tail_ptr;      // [B + 1]
  const int* tail_pieces;
  double* Lval;
  double* Uval;             // update matrices handed from child pieces to their parents
  double* y;                // forward-substituted rhs, elimination order [dim]
  int* fail;                // [B]
  const unsigned* fblob;    // front tables (front_plan.hpp): the blobs, and per launch-order piece where its blob is; nullptr: record plan only
  const FrontGrp* lfgrp;
  int flat_L;               // 1: the factor is written in flat form (multi right-hand-side kernels); 0: class-interleaved (LM loop)
  long long* dbg;           // SSLAM_CHOL_STAMPS: shader-clock totals per phase of workgroup 0 ([16] tail kernel, [16] per-depth kernels)
};

// Speculative damping trials of ONE small graph (B == 1): g2o's LM retries a rejected step with lambda * nu, nu * 2, ... up to ten times
// (SURVEY A.3), every retry a factorisation + solve + chi2 of the SAME linearisation -- and an optimize() call ends with ten rejected
// trials in a row.  The ten lambdas of an iteration are known when it starts, the trials do not depend on each other, and a graph of the
// orchestrator's size leaves the chip idle: all of them run side by side ("lanes": own L, update matrices, y, x, trial estimates, partial
// sums), then one thread replays the accept / reject sequence over their results in order.  Same arithmetic per trial, same decisions:
// bitwise the sequential result, an LM iteration in one round of launches instead of up to ten.
struct SpecLanes {
  int K = 0;                  // lanes (0: off)
  LmState* lm = nullptr;      // [K][B] per-lane copy of the graph's state (lambda of the lane, in_trial)
  double *Lval = nullptr, *Uval = nullptr, *y = nullptr, *x = nullptr, *pose_trial = nullptr, *lmk_trial = nullptr, *part_e = nullptr, *part_a = nullptr;
  int *fail = nullptr, *flow = nullptr;
  int* err = nullptr;         // ONE timeout flag for all lanes (the plan's own: d_flow + 3 * npiece) -- chol_flow_check reads a single word
  const double *pose_cur = nullptr, *lmk_cur = nullptr;   // the current estimates (V.pose / V.lmk)
  long long sL = 0, sU = 0, sy = 0, sx = 0, spose = 0, slmk = 0, spe = 0, spa = 0, sflow = 0;   // lane strides (elements)
  int g0 = 0, g1 = 0;         // workgroups of lane 0 / of every other lane in the (one-dimensional) grid of k_chol_flow
  int after = 0;              // lanes 1.. take part once `after` trials of the iteration have been rejected (0: always; 1: adaptive)
  int* ctl = nullptr;         // k_chol_spec_round: [0] rounds begun, [1] lanes at work in this round, [2] lanes finished in this round,
                              // [8 + k] end tickets of lane k, [8 + K + k] rounds lane k has run (its epoch)
};

struct CholPlan {
  CholView C{};
  SpecLanes spec{};
  std::vector<int> lvl_ptr, plv_ptr, plv_lds_f, plv_lds_b, plv_nt, plv_cls;
  std::vector<int> plv_lds_ff;  // front kernels (front_kernels.hpp): LDS doubles per launch; front: the per-depth launches run them
  int tail_lds_ff = 0;
  bool front = false;
  int tail_lds_f = 0, tail_lds_b = 0, tail_total = 0, nt_tail = 512, nt_ftail = 256, nt_bleaf = 0, nt_bmid = 0, nt_btail = 0, nt_leaf = 64, ustage = 0;
  std::vector<void*> allocs;
  DevArena* arena = nullptr;    // the owning batch's arena (single-graph handles), else hipMalloc
  int64_t lnz = 0, unz = 0;
  // active-graph compaction (LM endgame: a retry by a handful of graphs must not dispatch the pieces of all of them)
  std::vector<int> lp_graph;    // graph of every launch-order piece record
  std::vector<int> c_ptr;       // per launch: first entry of its compact index list in d_idx; back() = first tail entry; then the end
  int* d_idx = nullptr;         // [pieces of the active graphs, launch by launch | active graphs that have a tail]
  size_t idx_cap = 0;
  bool compact = false;
  double* d_multi_y = nullptr;  // scratch for multi-rhs solves
  double* d_multi_x = nullptr;
  int multi_cap = 0;
  // dependency-driven factorisation + solve in ONE launch (k_chol_flow): small batches only
  bool flow = false;            // the plan can run it (every piece has one parent piece; nt_leaf == nt_tail)
  int flow_grid = 0;            // persistent workgroups
  int spec_grid = 0;            // workgroups of a speculative round (all lanes)
  double flow_need = 1.0, spec_need = 1.0;   // the share of the device those grids occupy when resident (workgroups / what the device holds of them)
  int flow_first = 0;           // first launch-order piece of the single launch; the per-depth launches [0, flow_launch0) come before it
  int flow_launch0 = 0;
  int flow_epoch = 0;           // launches so far: the counters are never reset, a launch waits for epoch * (children)
  int lm_epoch = 0;             // launches that carried the LM halves of a trial (k_chol_flow, bit 2)
  int spec_epoch = 0;           // the same for the lanes' own counters (speculative trials)
  int2* d_dep = nullptr;        // per launch-order piece: {parent (launch order) or -1, children}
  int* d_flow = nullptr;        // [children done | backward done | forward done] per piece, then [0] error flag at 3 * npiece
  // marginals along the elimination-tree paths (k_chol_marginal_paths)
  std::vector<int> h_cparent;   // parent column in the elimination tree, -1: root
  std::vector<int> h_xoff_col;  // offset in the unknown vector (internal row order) -> column, -1 elsewhere
  int* d_mpath = nullptr;       // [path_ptr | dims | path columns] of a request
  double* d_mout = nullptr;     // [requests][36]
  size_t mpath_cap = 0, mout_cap = 0;
};

void chol_plan_free(CholPlan* p) {
  if (!p) return;
  if (p->C.dbg) {   // SSLAM_CHOL_STAMPS: where workgroup 0 of the factor kernels spent its shader clocks
    long long h[64];
    if (hipMemcpy(h, p->C.dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int k = 0; k < 2; ++k) {
        const long long* d = h + 32 + 8 * k;
        fprintf(stderr, "[chol-stamps] backward %s: pieces %lld levels %lld blocks %lld | clocks: blocks pass %lld column sums %lld levels %lld store %lld\n",
                k ? "depth kernels (wg 0)" : "tail (graph 0)", d[4], d[5], d[6], d[0], d[1], d[2], d[3]);
      }
      if (p->front) for (int k = 0; k < 3; ++k) {
        const long long* d = h + (k == 2 ? 48 : 16 * k);
        fprintf(stderr, "[front-stamps] %s: pieces %lld levels %lld U-tiles %lld target tiles %lld | clocks: blob %lld derive %lld gather %lld tiles %lld factor+rows %lld U %lld out %lld\n",
                k == 2 ? "mid pieces (wg 0)" : k ? "leaf groups (wg 0)" : "tail (graph 0)", d[8], d[9], d[10], d[11], d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
      }
      else for (int k = 0; k < 2; ++k) {
        const long long* d = h + 16 * k;
        fprintf(stderr, "[chol-stamps] %s: pieces %lld levels %lld U-items %lld int-items %lld | clocks: tables %lld gather %lld items %lld reduce %lld diag %lld rows %lld U %lld out %lld\n",
                k ? "depth kernels (wg 0)" : "tail (graph 0)", d[8], d[9], d[10], d[11], d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
      }
    }
  }
  for (void* a : p->allocs) (void)hipFree(a);
  if (p->d_idx) (void)hipFree(p->d_idx);
  if (p->d_multi_y) (void)hipFree(p->d_multi_y);
  if (p->d_multi_x) (void)hipFree(p->d_multi_x);
  if (p->d_mpath && !p->arena) (void)hipFree(p->d_mpath);
  if (p->d_mout && !p->arena) (void)hipFree(p->d_mout);
  delete p;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// broadcast lane K of every quad (DPP quad_perm: a VALU move, no LDS traffic)
template <int K>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true); }

struct alignas(16) D2 { double a, b; };
// one update S(i,j) -= L(i,k) L(j,k)^T (and rhs_j -= L(j,k) y_k on the diagonal) restricted to this lane's 3 x 3 tile.
// Ls / Ys: the piece's LDS mirror of L / y.  DK (width of the source column k) is a template parameter so that every offset is an
// instruction immediate; for DK = 6 the three rows of each operand are 16-byte aligned (blocks are stored at even offsets) and
// travel as 128-bit loads: 18 LDS loads + 6 for y feed 63 FMAs.
template <int DK>
__device__ __forceinline__ void tile_update_k(const double* __restrict__ Ls, const double* __restrict__ Ys, int ua, int ub, int ux,
                                              int tre, int tce, double (&acc)[9], double (&accy)[3]) {
  const double* A = Ls + ua + 3 * tre * DK;
  const double* B = Ls + ub + 3 * tce * DK;
  const double* yk = Ys + ux;
  double bb[3 * DK], yv[DK];
  if (DK == 6) {
    const D2* B2 = reinterpret_cast<const D2*>(B);
#pragma unroll
    for (int q = 0; q < 9; ++q) { const D2 v = B2[q]; bb[2 * q] = v.a; bb[2 * q + 1] = v.b; }
  } else {
#pragma unroll
    for (int q = 0; q < 3 * DK; ++q) bb[q] = B[q];
  }
#pragma unroll
  for (int q = 0; q < DK; ++q) yv[q] = yk[q];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {   // one row of the A operand at a time: 30 live doubles instead of 42
    double a[DK];
    if (DK == 6) {
      const D2* A2 = reinterpret_cast<const D2*>(A + rr * DK);
#pragma unroll
      for (int q = 0; q < 3; ++q) { const D2 v = A2[q]; a[2 * q] = v.a; a[2 * q + 1] = v.b; }
    } else {
#pragma unroll
      for (int q = 0; q < DK; ++q) a[q] = A[rr * DK + q];
    }
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      double v = acc[rr * 3 + cc];
#pragma unroll
      for (int q = 0; q < DK; ++q) v += a[q] * bb[cc * DK + q];
      acc[rr * 3 + cc] = v;
    }
    double w = accy[rr];
#pragma unroll
    for (int q = 0; q < DK; ++q) w += a[q] * yv[q];
    accy[rr] = w;
  }
}
__device__ __forceinline__ void tile_update(const double* __restrict__ Ls, const double* __restrict__ Ys, int ua, int ub, int ux, unsigned pk,
                                            int tre, int tce, double (&acc)[9], double (&accy)[3]) {
  if (pk & kUpdDk6) tile_update_k<6>(Ls, Ys, ua, ub, ux, tre, tce, acc, accy);
  else tile_update_k<3>(Ls, Ys, ua, ub, ux, tre, tce, acc, accy);
}

// Work items [it_begin, it_end): four lanes per item (lane = 3 x 3 tile (tr, tc) of the target block).  The item's update
// records are fetched eight at a time, two per lane, and handed round the quad with DPP broadcasts: one memory latency per
// eight updates instead of one per update.  A sole item subtracts its tile from the LDS-resident target in place; the items
// of a split list park their tiles in `part` (summed in item order by reduce_multi).
template <int NT>
__device__ __forceinline__ void run_items(const ItemMeta* items, int it_begin, int it_end, const UpdMeta* __restrict__ upd,
                                          const double* __restrict__ Ls, const double* __restrict__ Ys, int lofs, int yofs,
                                          double* smL, double* smY, double* part, int tid) {
  const int lq = tid & 3, tr = lq >> 1, tc = lq & 1;
  for (int it = it_begin + (tid >> 2); it < it_end; it += NT / 4) {
    const ItemMeta im = items[it];
    const int n = im.n;
    double acc[9], accy[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) accy[q] = 0;
    int di = 6, dj = 6, tre = 0, tce = 0;
    bool diag = false;
    for (int k0 = 0; k0 < n; k0 += 8) {
      const UpdMeta r0 = upd[im.u0 + min(k0 + lq, n - 1)];
      const UpdMeta r1 = upd[im.u0 + min(k0 + 4 + lq, n - 1)];
      if (k0 == 0) {
        const unsigned pk0 = (unsigned)quad_bcast<0>((int)r0.xk);
        di = (pk0 & kUpdDi6) ? 6 : 3; dj = (pk0 & kUpdDj6) ? 6 : 3; diag = pk0 & kUpdDiag;
        tre = 3 * tr < di ? tr : 0; tce = 3 * tc < dj ? tc : 0;   // idle lanes shadow tile (0, 0): valid addresses
      }
#define SSLAM_STEP(KK, R)                                                                                        \
  if (k0 + KK < n) {                                                                                             \
    const UpdMeta q_{(unsigned)quad_bcast<(KK) & 3>((int)R.ab), (unsigned)quad_bcast<(KK) & 3>((int)R.xk)};        \
    tile_update(Ls, Ys, upd_ua(q_), upd_ub(q_), upd_yk(q_), q_.xk, tre, tce, acc, accy);                          \
  }
      SSLAM_STEP(0, r0) SSLAM_STEP(1, r0) SSLAM_STEP(2, r0) SSLAM_STEP(3, r0)
      SSLAM_STEP(4, r1) SSLAM_STEP(5, r1) SSLAM_STEP(6, r1) SSLAM_STEP(7, r1)
#undef SSLAM_STEP
    }
    if (3 * tr < di && 3 * tc < dj) {
      if (im.flags & kItemSole) {
        double* o = smL + im.tloff;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] -= acc[rr * 3 + cc];
        if (diag && tc == 0) {
          double* oy = smY + (im.flags >> kItemYShift);
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) oy[3 * tr + rr] -= accy[rr];
        }
      } else {
        double* o = part + ((im.flags >> kItemSlotShift) & kItemSlotMask) * kItemDoubles;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
          if (diag && tc == 0) o[36 + 3 * tr + rr] = accy[rr];
        }
      }
    }
  }
}

// targets whose list was split: sum the partial tiles in item order (one wave per target, lane = entry)
__device__ __forceinline__ void reduce_multi(const MbMeta* __restrict__ mbs, int m0, int m1, double* smL, double* smY, const double* part,
                                             int wave, int lane, int nw) {
  const int ry = lane - 40;
  for (int m = m0 + wave; m < m1; m += nw) {
    const MbMeta mm = mbs[m];
    const int di = mm.info & 15, dj = (mm.info >> 4) & 15;
    const bool diag = mm.info & kBlkDiag;
    const double* p = part + mm.ps0 * kItemDoubles;
    if (lane < di * dj) {
      double v = 0;
      for (int q = 0; q < mm.n; ++q) v += p[q * kItemDoubles + lane];
      smL[mm.tloff + lane] -= v;
    } else if (diag && ry >= 0 && ry < dj) {
      double v = 0;
      for (int q = 0; q < mm.n; ++q) v += p[q * kItemDoubles + 36 + ry];
      smY[(mm.info >> 12) + ry] -= v;
    }
  }
}

// Update-matrix items [it_begin, it_end) of a piece: the block U(a,b) = sum of the piece's own updates L(a,k) L(b,k)^T (sources in
// LDS) + the blocks its children handed up for the same pair; a diagonal block carries the rhs part sum_k L(a,k) y_k along.
// A sole item finishes its block and writes it to HBM; the items of a split list park their tiles in `part` (reduce_umulti).
template <int NT>
__device__ __forceinline__ void run_uitems(const UItem* __restrict__ items, int it_begin, int it_end, const UpdMeta* __restrict__ upd,
                                           const AsmSrc* __restrict__ usrc, const double* __restrict__ smL, const double* __restrict__ smY,
                                           int lofs, int yofs, double* __restrict__ U, double* part, int tid) {
  const int lq = tid & 3, tr = lq >> 1, tc = lq & 1;
  // One item per quad and pass.  A pass used to be three dependent trips to HBM (item record -> its first child-source record -> the
  // child's block) in front of the arithmetic, and this phase was half of a leaf piece's clocks (SSLAM_CHOL_STAMPS): the record of the
  // NEXT pass's item and its first source are now fetched one pass ahead, so that a pass starts with everything it needs to issue its
  // block loads at once (512 factorisations 10.97 -> 10.48 ms).  (Fetching the next item's first update records ahead as well costs 16
  // spilled VGPRs under the 128-register cap of the leaf kernel and was slower: 12.0 ms.)
  int it = it_begin + (tid >> 2);
  UItem nxt = items[min(it, max(it_end - 1, it_begin))];
  AsmSrc nsrc = ((nxt.flags & kItemSole) && nxt.ns > 0) ? usrc[nxt.s0] : AsmSrc{0, -1};
  for (; it < it_end; it += NT / 4) {
    const UItem im = nxt;
    const AsmSrc src0 = nsrc;
    nxt = items[min(it + NT / 4, it_end - 1)];
    const int n = im.n;
    const int di = (im.flags & kUItemDi6) ? 6 : 3, dj = (im.flags & kUItemDj6) ? 6 : 3;
    const bool diag = im.flags & kUItemDiag;
    const int tre = 3 * tr < di ? tr : 0, tce = 3 * tc < dj ? tc : 0;
    double acc[9], accy[3];
    // the first child block of this tile: its loads travel while the own updates are computed out of LDS
    const bool sole = im.flags & kItemSole;
    {
      const double* o = U + src0.uoff;
      const bool on = sole && im.ns > 0;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) { const double v = o[(3 * tre + rr) * dj + 3 * tce + cc]; acc[rr * 3 + cc] = on ? v : 0.0; }
      const double* oy = U + (src0.uyoff >= 0 ? src0.uyoff : 0);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { const double v = oy[3 * tre + rr]; accy[rr] = (on && diag && src0.uyoff >= 0) ? v : 0.0; }
    }
    nsrc = ((nxt.flags & kItemSole) && nxt.ns > 0) ? usrc[nxt.s0] : AsmSrc{0, -1};
    for (int k0 = 0; k0 < n; k0 += 8) {
      const UpdMeta r0 = upd[im.u0 + min(k0 + lq, n - 1)];
      const UpdMeta r1 = upd[im.u0 + min(k0 + 4 + lq, n - 1)];
#define SSLAM_STEP(KK, R)                                                                                        \
  if (k0 + KK < n) {                                                                                             \
    const UpdMeta q_{(unsigned)quad_bcast<(KK) & 3>((int)R.ab), (unsigned)quad_bcast<(KK) & 3>((int)R.xk)};        \
    tile_update(smL, smY, upd_ua(q_), upd_ub(q_), upd_yk(q_), q_.xk, tre, tce, acc, accy);                        \
  }
      SSLAM_STEP(0, r0) SSLAM_STEP(1, r0) SSLAM_STEP(2, r0) SSLAM_STEP(3, r0)
      SSLAM_STEP(4, r1) SSLAM_STEP(5, r1) SSLAM_STEP(6, r1) SSLAM_STEP(7, r1)
#undef SSLAM_STEP
    }
    if (3 * tr < di && 3 * tc < dj) {
      if (sole) {
        for (int s2 = 1; s2 < im.ns; ++s2) {
          const AsmSrc src = usrc[im.s0 + s2];
          const double* o = U + src.uoff;
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) acc[rr * 3 + cc] += o[(3 * tr + rr) * dj + 3 * tc + cc];
          if (diag && tc == 0 && src.uyoff >= 0) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) accy[rr] += U[src.uyoff + 3 * tr + rr];
          }
        }
        double* o = U + im.uoff;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
        if (diag && tc == 0) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) U[im.uyoff + 3 * tr + rr] = accy[rr];
        }
      } else {
        double* o = part + ((im.flags >> kItemSlotShift) & kItemSlotMask) * kItemDoubles;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
          if (diag && tc == 0) o[36 + 3 * tr + rr] = accy[rr];
        }
      }
    }
  }
}

// The sole U items of a piece, one 3 x 3 TILE per lane (round 4).  The items arrive ordered by tile count (PieceMeta.nu4 / nu2 / nu1: 6 x 6
// blocks, 6 x 3 and 3 x 6, 3 x 3), so a lane finds its (item, tile) by arithmetic.  With a quad per item a 3 x 3 block kept one lane of
// four busy, and landmark-landmark blocks are most of a leaf piece's update matrix: 31 % lane utilisation, five passes over the item
// list where two do -- and every pass is a chain of dependent trips to HBM.  Same arithmetic per tile as run_uitems (tile_update, then the
// children's blocks in list order): same results.  The next tile's item record and first child source are fetched one pass ahead.
template <int NT>
__device__ __forceinline__ void run_utiles(const UItem* __restrict__ items, const int n4, const int n2, const int n1, const UpdMeta* __restrict__ upd,
                                           const AsmSrc* __restrict__ usrc, const double* __restrict__ smL, const double* __restrict__ smY,
                                           int lofs, int yofs, double* __restrict__ U, int tid) {
  const int T4 = 4 * n4, T2 = T4 + 2 * n2, T = T2 + n1;
  auto decode = [&](int t, int& q) -> int {
    if (t < T4) { q = t & 3; return t >> 2; }
    if (t < T2) { const int u = t - T4; q = u & 1; return n4 + (u >> 1); }
    q = 0;
    return n4 + n2 + (t - T2);
  };
  int t = tid, nq = 0;
  UItem nxt = items[decode(min(t, T - 1), nq)];
  AsmSrc nsrc = nxt.ns > 0 ? usrc[nxt.s0] : AsmSrc{0, -1};
  for (; t < T; t += NT) {
    const UItem im = nxt;
    const AsmSrc src0 = nsrc;
    const int qq = nq;
    nxt = items[decode(min(t + NT, T - 1), nq)];
    const int n = im.n;
    const int di = (im.flags & kUItemDi6) ? 6 : 3, dj = (im.flags & kUItemDj6) ? 6 : 3;
    const bool diag = im.flags & kUItemDiag;
    int tr, tc;
    if (di == 6 && dj == 6) { tr = qq >> 1; tc = qq & 1; }
    else if (di == 6) { tr = qq; tc = 0; }
    else { tr = 0; tc = qq; }
    double acc[9], accy[3];
    {
      const bool on = im.ns > 0;
      const double* o = U + src0.uoff;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) { const double v = o[(3 * tr + rr) * dj + 3 * tc + cc]; acc[rr * 3 + cc] = on ? v : 0.0; }
      const double* oy = U + (src0.uyoff >= 0 ? src0.uyoff : 0);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { const double v = oy[3 * tr + rr]; accy[rr] = (on && diag && src0.uyoff >= 0) ? v : 0.0; }
    }
    nsrc = nxt.ns > 0 ? usrc[nxt.s0] : AsmSrc{0, -1};
    for (int k0 = 0; k0 < n; k0 += 2) {   // the item's own updates out of LDS, records two at a time
      const UpdMeta r0 = upd[im.u0 + k0], r1 = upd[im.u0 + min(k0 + 1, n - 1)];
      tile_update(smL, smY, upd_ua(r0), upd_ub(r0), upd_yk(r0), r0.xk, tr, tc, acc, accy);
      if (k0 + 1 < n) tile_update(smL, smY, upd_ua(r1), upd_ub(r1), upd_yk(r1), r1.xk, tr, tc, acc, accy);
    }
    for (int s2 = 1; s2 < im.ns; ++s2) {
      const AsmSrc src = usrc[im.s0 + s2];
      const double* o = U + src.uoff;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) acc[rr * 3 + cc] += o[(3 * tr + rr) * dj + 3 * tc + cc];
      if (diag && tc == 0 && src.uyoff >= 0) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) accy[rr] += U[src.uyoff + 3 * tr + rr];
      }
    }
    double* o = U + im.uoff;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
    if (diag && tc == 0) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) U[im.uyoff + 3 * tr + rr] = accy[rr];
    }
  }
}

// U blocks whose own list was split: partial tiles in item order + the children's blocks -> HBM (one wave per block, lane = entry)
__device__ __forceinline__ void reduce_umulti(const UMb* __restrict__ mbs, int m0, int m1, const AsmSrc* __restrict__ usrc, double* __restrict__ U,
                                              const double* part, int wave, int lane, int nw) {
  const int ry = lane - 40;
  for (int m = m0 + wave; m < m1; m += nw) {
    const UMb mm = mbs[m];
    const int di = mm.info & 15, dj = (mm.info >> 4) & 15;
    const bool diag = mm.info & kBlkDiag;
    const double* p = part + mm.ps0 * kItemDoubles;
    if (lane < di * dj) {
      double v = 0;
      for (int q = 0; q < mm.n; ++q) v += p[q * kItemDoubles + lane];
      for (int s2 = 0; s2 < mm.ns; ++s2) v += U[usrc[mm.s0 + s2].uoff + lane];
      U[mm.uoff + lane] = v;
    } else if (diag && ry >= 0 && ry < dj) {
      double v = 0;
      for (int q = 0; q < mm.n; ++q) v += p[q * kItemDoubles + 36 + ry];
      for (int s2 = 0; s2 < mm.ns; ++s2) { const int uy = usrc[mm.s0 + s2].uyoff; if (uy >= 0) v += U[uy + ry]; }
      U[mm.uyoff + ry] = v;
    }
  }
}

// diagonal block of one column, one thread: L_jj = chol(S_jj) in place (strict upper = 0), reciprocal pivots to inv,
// y_j = L_jj^-1 rhs_j in place.  Right-looking form: after a pivot only one multiply and one FMA lie between it and the next
// pivot (the trailing updates are independent of one another), so the thread's latency is ~D x (rsqrt + 2 ops), not ~D^2 FMAs.
// Pivots are inverted once (rsqrt) and multiplied from then on.
template <int D>
__device__ __forceinline__ bool diag_factor(double* Ljj, double* yj, double* inv_out) {
  double a[D * D], t[D], inv[D];
#pragma unroll
  for (int q = 0; q < D * D; ++q) a[q] = Ljj[q];
#pragma unroll
  for (int q = 0; q < D; ++q) t[q] = yj[q];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double d = a[c * D + c];
    if (!(d > 0)) { ok = false; d = 1.0; }
    const double id = rsqrt(d);
    inv[c] = id;
    a[c * D + c] = d * id;
    t[c] *= id;
#pragma unroll
    for (int r = c + 1; r < D; ++r) a[r * D + c] *= id;
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
#pragma unroll
      for (int c2 = c + 1; c2 <= r; ++c2) a[r * D + c2] -= a[r * D + c] * a[c2 * D + c];
      t[r] -= a[r * D + c] * t[c];
    }
#pragma unroll
    for (int s2 = c + 1; s2 < D; ++s2) a[c * D + s2] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < D * D; ++q) Ljj[q] = a[q];
#pragma unroll
  for (int q = 0; q < D; ++q) { yj[q] = t[q]; inv_out[q] = inv[q]; }
  return ok;
}

// one row of an off-diagonal block of a factored column, one thread: x L_jj^T = v in place
template <int D>
__device__ __forceinline__ void row_solve(double* v, const double* Ljj, const double* invp) {
  double x[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double w = v[c];
#pragma unroll
    for (int s2 = 0; s2 < c; ++s2) w -= x[s2] * Ljj[c * D + s2];
    x[c] = w * invp[c];
  }
#pragma unroll
  for (int c = 0; c < D; ++c) v[c] = x[c];
}

// ------------------------------------------------------------------------------------------------
// factorisation of one piece by one workgroup
//   S = A + lambda I - sum_k L(:,k) L(j,k)^T,  L(j,j) = chol(S(j,j)),  L(i,j) = S(i,j) L(j,j)^-T,
//   y(j) = L(j,j)^-1 (b(j) - sum_k L(j,k) y(k))
// LDS: [L of the piece | y | reciprocal pivots | block table | column table | internal items | partial tiles]
// ------------------------------------------------------------------------------------------------
#define SSLAM_STAMP(k)                                                          \
  if (dbg) {                                                                    \
    const long long now_ = clock64();                                           \
    if (threadIdx.x == 0) dbg[k] += now_ - tprev;                               \
    tprev = now_;                                                               \
  }
// DEFER (k_chol_flow): the piece's tables and its part of H are fetched BEFORE the wait for the child pieces (they do not depend on
// them); only the children's update-matrix blocks are read after it -- two dependent round trips less on the critical path of a piece.
// The sums are those of the one-pass gather, in its order: (H + lambda I) first, then the sources one after the other.
__device__ __forceinline__ bool flow_wait(const int* p, int target, int* err, int* fail = nullptr);
template <int NT, bool USTAGE, bool RIGHT, bool DEFER = false>
__device__ __forceinline__ void chol_piece(const BatchView& V, const CholView& C, const PieceMeta pm, double* sm, long long* dbg,
                                           const int* wait_p = nullptr, int wait_target = 0, int* wait_err = nullptr) {
  constexpr int NW = NT / 64;
  long long tprev = dbg ? clock64() : 0;
  ILevel* s_lv = reinterpret_cast<ILevel*>(sm);              // the piece's level records first (32 bytes each): no static table, the LDS a
  const int tid = threadIdx.x;                               // piece reserves is what it uses (residency is LDS-bound)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = pm.graph;
  const double lambda = V.lm[g].lambda;
  const int Lp = (pm.lsize + 1) & ~1, Yp = (pm.ysize + 1) & ~1;
  double* smL = sm + 4 * pm.nilv;
  double* smY = smL + Lp;
  double* smInv = smY + Yp;
  BlkMeta* sBlk = reinterpret_cast<BlkMeta*>(smInv + Yp);                 // the piece's block records
  int4* sCol = reinterpret_cast<int4*>(sBlk + pm.nb);                      // {L offset of the diagonal block, dim, y offset, -} (piece-local)
  // internal updates: target-major items + records + multi-blocks, or (RIGHT) one {first record, count} per column + source-major records
  ItemMeta* sItem = reinterpret_cast<ItemMeta*>(sCol + pm.nc);
  UpdMeta* sUpd = reinterpret_cast<UpdMeta*>(sItem + (RIGHT ? (pm.nc + 1) / 2 : pm.nit_i));
  MbMeta* sMb = reinterpret_cast<MbMeta*>(sUpd + ((pm.nu_i + 1) & ~1));   // (8-byte records: an even count keeps what follows 16-byte aligned)
  AsmSrc* sAsm = reinterpret_cast<AsmSrc*>(sMb + (RIGHT ? 0 : pm.nimb));   // child update-matrix blocks to absorb
  RCol* sRcol = reinterpret_cast<RCol*>(sItem);
  // update-matrix records: staged in LDS by the per-depth kernels (USTAGE), read from HBM by the tail
  UItem* sUItem = reinterpret_cast<UItem*>(sAsm + pm.nas + (pm.nas & 1));
  UMb* sUMb = reinterpret_cast<UMb*>(sUItem + (USTAGE ? pm.nuit : 0));
  UpdMeta* sUUpd = reinterpret_cast<UpdMeta*>(sUMb + (USTAGE ? pm.numb : 0));
  AsmSrc* sUSrc = reinterpret_cast<AsmSrc*>(sUUpd + (USTAGE ? ((pm.nuu + 1) & ~1) : 0));
  double* part = reinterpret_cast<double*>(sUSrc + (USTAGE ? pm.nus + (pm.nus & 1) : 0));
  const double* __restrict__ H = V.Hpp_diag;
  const double* __restrict__ U = C.Uval;
  const int ry = lane - 40;
  // ---- 0. every table of the piece -> LDS in ONE round trip: all loads are issued (unconditionally, clamped indices) before the
  //         first store, then the stores; only tables longer than their unroll x NT go round again.
#define SSLAM_LD(T, name, src, n, UNR)                                                     \
  T name##_r[UNR];                                                                         \
  _Pragma("unroll") for (int k_ = 0; k_ < UNR; ++k_) name##_r[k_] = (src)[min(tid + k_ * NT, max((n)-1, 0))];
#define SSLAM_ST(name, dst, src, n, UNR)                                                   \
  _Pragma("unroll") for (int k_ = 0; k_ < UNR; ++k_) if (tid + k_ * NT < (n)) (dst)[tid + k_ * NT] = name##_r[k_]; \
  for (int i_ = tid + UNR * NT; i_ < (n); i_ += NT) (dst)[i_] = (src)[i_];
  {
    SSLAM_LD(ILevel, t_lv, C.ilv + pm.ilv0, pm.nilv, 1)
    SSLAM_LD(BlkMeta, t_blk, C.blk + pm.b0, pm.nb, 2)
    SSLAM_LD(ItemMeta, t_item, C.item + pm.iit0, RIGHT ? 0 : pm.nit_i, 2)
    SSLAM_LD(UpdMeta, t_upd, (RIGHT ? C.rupd + pm.pad3 : C.upd + pm.iu0), pm.nu_i, 2)
    SSLAM_LD(MbMeta, t_mb, C.mb + pm.imb0, RIGHT ? 0 : pm.nimb, 1)
    SSLAM_LD(RCol, t_rcol, C.rcol + pm.c0, RIGHT ? pm.nc : 0, 1)
    SSLAM_LD(AsmSrc, t_asm, C.asrc + pm.as0, pm.nas, 2)
    SSLAM_LD(ColMeta, t_col, C.col + pm.c0, pm.nc, 1)
    SSLAM_LD(UItem, t_uit, C.uitem + pm.uit0, USTAGE ? pm.nuit : 0, 2)
    SSLAM_LD(UMb, t_umb, C.umb + pm.umb0, USTAGE ? pm.numb : 0, 1)
    SSLAM_LD(UpdMeta, t_uupd, C.upd + pm.uu0, USTAGE ? pm.nuu : 0, 2)
    SSLAM_LD(AsmSrc, t_usrc, C.usrc + pm.us0, USTAGE ? pm.nus : 0, 2)
    SSLAM_ST(t_lv, s_lv, C.ilv + pm.ilv0, pm.nilv, 1)
    SSLAM_ST(t_blk, sBlk, C.blk + pm.b0, pm.nb, 2)
    SSLAM_ST(t_item, sItem, C.item + pm.iit0, RIGHT ? 0 : pm.nit_i, 2)
    SSLAM_ST(t_upd, sUpd, (RIGHT ? C.rupd + pm.pad3 : C.upd + pm.iu0), pm.nu_i, 2)
    SSLAM_ST(t_mb, sMb, C.mb + pm.imb0, RIGHT ? 0 : pm.nimb, 1)
    SSLAM_ST(t_rcol, sRcol, C.rcol + pm.c0, RIGHT ? pm.nc : 0, 1)
    SSLAM_ST(t_asm, sAsm, C.asrc + pm.as0, pm.nas, 2)
    if (tid < pm.nc) sCol[tid] = make_int4(t_col_r[0].base - pm.lbase, t_col_r[0].dim, t_col_r[0].yoff - pm.y0, 0);
    for (int c = tid + NT; c < pm.nc; c += NT) {
      const ColMeta cm = C.col[pm.c0 + c];
      sCol[c] = make_int4(cm.base - pm.lbase, cm.dim, cm.yoff - pm.y0, 0);
    }
    if (USTAGE) {
      SSLAM_ST(t_uit, sUItem, C.uitem + pm.uit0, pm.nuit, 2)
      SSLAM_ST(t_umb, sUMb, C.umb + pm.umb0, pm.numb, 1)
      SSLAM_ST(t_uupd, sUUpd, C.upd + pm.uu0, pm.nuu, 2)
      SSLAM_ST(t_usrc, sUSrc, C.usrc + pm.us0, pm.nus, 2)
    }
  }
#undef SSLAM_LD
#undef SSLAM_ST
  __syncthreads();
  SSLAM_STAMP(0)
  // ---- 1. gather: A(:, piece) + lambda I and the rhs, minus what the child pieces left for these blocks (their update matrices)
  //         -> LDS.  One thread per (block, row): a handful of instructions per row (these kernels are bound by instruction issue,
  //         not by bytes).  kG rows per thread are in flight together, H row and first child row side by side; a row's sources
  //         are summed by its own thread -> no conflicts, fixed order.
  {
    // KG rows per thread in flight together; SRC: the rows' first child block rides along with the H row.  A piece without child blocks (every
    // group of the bottom launch: a quarter of all groups; and the first pass of DEFER) loads the H rows only and keeps five per thread in
    // flight: the 600 rows of a 100-block group are one round trip for 128 threads instead of three (round 5; the 64-thread pieces of
    // rounds 2-4 had 96 rows and were one trip at two per thread).
    auto gather = [&](auto KGc, auto SRCc) {
      constexpr int KG = decltype(KGc)::value;
      constexpr bool SRC = decltype(SRCc)::value;
      const int nrow = pm.nb * 6;
      for (int t0 = tid; t0 < nrow; t0 += NT * KG) {
        double v[KG][6], w[SRC ? KG : 1][6], rhsv[KG], uyv[SRC ? KG : 1];
#pragma unroll
        for (int g2 = 0; g2 < KG; ++g2) {
          const int t = min(t0 + g2 * NT, nrow - 1);
          const int b = t / 6, row = t - 6 * b;
          const BlkMeta bm = sBlk[b];
          const int di = bm.info & 15, dj = (bm.info >> 4) & 15, nas = SRC ? (bm.info >> kBlkNasShift) & 255 : 0;
          const int rw = min(row, di - 1);                                   // idle threads shadow the last row: valid addresses
          const double* ph = H + max(bm.src, 0) + ((bm.info & kBlkFmt) ? rw : rw * dj);
          const int st = (bm.info & kBlkFmt) ? di : 1;
          const AsmSrc as = nas > 0 ? sAsm[bm.as0] : AsmSrc{0, -1};
          const double* pu = U + as.uoff + rw * dj;
          if (dj == 6 && !(bm.info & kBlkFmt)) {   // 48 contiguous, 16-byte aligned bytes: three wide loads each
            const D2* ph2 = reinterpret_cast<const D2*>(ph);
            const D2* pu2 = reinterpret_cast<const D2*>(pu);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const D2 a2 = ph2[c]; v[g2][2 * c] = a2.a; v[g2][2 * c + 1] = a2.b;
              if (SRC) { const D2 b2 = pu2[c]; w[g2][2 * c] = b2.a; w[g2][2 * c + 1] = b2.b; }
            }
          } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) { const int cc = min(c, dj - 1); v[g2][c] = ph[cc * st]; if (SRC) w[g2][c] = pu[cc]; }
          }
          const bool dg = bm.info & kBlkDiag;                                 // non-diagonal rows all read element 0: one request per wave
          rhsv[g2] = V.bvec[dg ? bm.xoff_row + rw : 0];
          if (SRC) uyv[g2] = U[(dg && as.uyoff >= 0) ? as.uyoff + rw : 0];
        }
#pragma unroll
        for (int g2 = 0; g2 < KG; ++g2) {
          const int t = t0 + g2 * NT;
          if (t >= nrow) continue;
          const int b = t / 6, row = t - 6 * b;
          const BlkMeta bm = sBlk[b];
          const int di = bm.info & 15, dj = (bm.info >> 4) & 15, nas = SRC ? (bm.info >> kBlkNasShift) & 255 : 0;
          if (row >= di) continue;
          const bool diag = bm.info & kBlkDiag;
          double vy = diag ? rhsv[g2] : 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) v[g2][c] = (bm.src >= 0 ? v[g2][c] : 0.0) + ((diag && c == row) ? lambda : 0.0) - ((SRC && nas > 0) ? w[SRC ? g2 : 0][c] : 0.0);
          if (SRC && nas > 0) {
            const AsmSrc as0 = sAsm[bm.as0];
            if (diag && as0.uyoff >= 0) vy -= uyv[SRC ? g2 : 0];
            for (int s2 = 1; s2 < nas; ++s2) {
              const AsmSrc as = sAsm[bm.as0 + s2];
              const double* pu = U + as.uoff + row * dj;
#pragma unroll
              for (int c = 0; c < 6; ++c) if (c < dj) v[g2][c] -= pu[c];
              if (diag && as.uyoff >= 0) vy -= U[as.uyoff + row];
            }
          }
          double* o = smL + (bm.off - pm.lbase) + row * dj;
#pragma unroll
          for (int c = 0; c < 6; ++c) if (c < dj) o[c] = v[g2][c];
          if (diag) smY[bm.colyoff - pm.y0 + row] = vy;
        }
      }
    };
    if (DEFER || pm.nas == 0) gather(std::integral_constant<int, 5>{}, std::false_type{});
    else gather(std::integral_constant<int, 3>{}, std::true_type{});
  }
  if (DEFER) {
    // ---- 1b. the children's update-matrix blocks, once they are there
    if (wait_p) {
      __syncthreads();
      if (tid == 0) flow_wait(wait_p, wait_target, wait_err, C.fail + g);
    }
    __syncthreads();
    if (pm.nas > 0) {
      const int nrow = pm.nb * 6;
      for (int t = tid; t < nrow; t += NT) {
        const int b = t / 6, row = t - 6 * b;
        const BlkMeta bm = sBlk[b];
        const int di = bm.info & 15, dj = (bm.info >> 4) & 15, nas = (bm.info >> kBlkNasShift) & 255;
        if (row >= di || nas == 0) continue;
        const bool diag = bm.info & kBlkDiag;
        double* o = smL + (bm.off - pm.lbase) + row * dj;
        double* oy = smY + (bm.colyoff - pm.y0 + row);
        double v[6], vy = diag ? *oy : 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = c < dj ? o[c] : 0.0;
        for (int s2 = 0; s2 < nas; ++s2) {
          const AsmSrc as = sAsm[bm.as0 + s2];
          const double* pu = U + as.uoff + row * dj;
#pragma unroll
          for (int c = 0; c < 6; ++c) if (c < dj) v[c] -= pu[c];
          if (diag && as.uyoff >= 0) vy -= U[as.uyoff + row];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) if (c < dj) o[c] = v[c];
        if (diag) *oy = vy;
      }
    }
  }
  __syncthreads();
  SSLAM_STAMP(1)
  // ---- 2. the levels inside the piece, everything in LDS
  for (int il = 0; il < pm.nilv; ++il) {
    const ILevel lv = s_lv[il];
    if (!RIGHT && lv.it1 > lv.it0) {
      run_items<NT>(sItem, lv.it0, lv.it1, sUpd, smL, smY, pm.lbase, pm.y0, smL, smY, part, tid);
      __syncthreads();
      SSLAM_STAMP(2)
      if (lv.mb1 > lv.mb0) {
        reduce_multi(sMb, lv.mb0, lv.mb1, smL, smY, part, wave, lane, NW);
        __syncthreads();
        SSLAM_STAMP(3)
      }
    }
    for (int c = lv.c0 - pm.c0 + tid; c < lv.c1 - pm.c0; c += NT) {
      const int4 cm = sCol[c];
      bool ok;
      if (cm.y == 6) ok = diag_factor<6>(smL + cm.x, smY + cm.z, smInv + cm.z);
      else ok = diag_factor<3>(smL + cm.x, smY + cm.z, smInv + cm.z);
      if (!ok) C.fail[g] = 1;
    }
    __syncthreads();
    SSLAM_STAMP(4)
    for (int t = tid; t < (lv.b1 - lv.b0) * 6; t += NT) {   // thread = (block, row)
      const int b = t / 6, row = t - 6 * b;
      const BlkMeta bm = sBlk[lv.b0 - pm.b0 + b];
      const int di = bm.info & 15, dj = (bm.info >> 4) & 15;
      if ((bm.info & kBlkDiag) || row >= di) continue;
      if (dj == 6) row_solve<6>(smL + (bm.off - pm.lbase) + row * 6, smL + (bm.coldiag - pm.lbase), smInv + (bm.colyoff - pm.y0));
      else row_solve<3>(smL + (bm.off - pm.lbase) + row * 3, smL + (bm.coldiag - pm.lbase), smInv + (bm.colyoff - pm.y0));
    }
    __syncthreads();
    SSLAM_STAMP(5)
    if (RIGHT) {
      // the finished columns update every later block of the piece, one column per round (two columns of a level may meet in a target):
      // four lanes per tile, one tile update deep
      const int lq = tid & 3, tr = lq >> 1, tc = lq & 1;
      for (int c = lv.c0 - pm.c0; c < lv.c1 - pm.c0; ++c) {
        const RCol rc = sRcol[c];
        const int yk = sCol[c].z;
        for (int it = tid >> 2; it < rc.n; it += NT / 4) {
          const UpdMeta r = sUpd[rc.u0 + it];
          const int di = (r.xk & kUpdDi6) ? 6 : 3, dj = (r.xk & kUpdDj6) ? 6 : 3;
          const int tre = 3 * tr < di ? tr : 0, tce = 3 * tc < dj ? tc : 0;   // idle lanes shadow tile (0, 0): valid addresses
          double acc[9], accy[3];
#pragma unroll
          for (int q = 0; q < 9; ++q) acc[q] = 0;
#pragma unroll
          for (int q = 0; q < 3; ++q) accy[q] = 0;
          tile_update(smL, smY, upd_ua(r), upd_ub(r), yk, r.xk, tre, tce, acc, accy);
          if (3 * tr < di && 3 * tc < dj) {
            double* o = smL + upd_rt(r);
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] -= acc[rr * 3 + cc];
            if ((r.xk & kUpdDiag) && tc == 0) {
              double* oy = smY + upd_ry(r);
#pragma unroll
              for (int rr = 0; rr < 3; ++rr) oy[3 * tr + rr] -= accy[rr];
            }
          }
        }
        __syncthreads();
      }
      SSLAM_STAMP(2)
    }
  }
  // ---- 3. the update matrix over the rows above the piece: own updates out of LDS + the children's blocks -> HBM
  if (pm.nuit > 0) {
    // sole items by tiles (run_utiles), the items of split lists by quads into their partial slots (run_uitems)
    const int nsole = pm.nu4 + pm.nu2 + pm.nu1;
    if (USTAGE) {
      if (nsole > 0) run_utiles<NT>(sUItem, pm.nu4, pm.nu2, pm.nu1, sUUpd, sUSrc, smL, smY, pm.lbase, pm.y0, C.Uval, tid);
      if (pm.nuit > nsole) run_uitems<NT>(sUItem, nsole, pm.nuit, sUUpd, sUSrc, smL, smY, pm.lbase, pm.y0, C.Uval, part, tid);
    } else {
      if (nsole > 0) run_utiles<NT>(C.uitem + pm.uit0, pm.nu4, pm.nu2, pm.nu1, C.upd + pm.uu0, C.usrc + pm.us0, smL, smY, pm.lbase, pm.y0, C.Uval, tid);
      if (pm.nuit > nsole) run_uitems<NT>(C.uitem + pm.uit0, nsole, pm.nuit, C.upd + pm.uu0, C.usrc + pm.us0, smL, smY, pm.lbase, pm.y0, C.Uval, part, tid);
    }
    if (pm.numb > 0) {
      __syncthreads();
      if (USTAGE) reduce_umulti(sUMb, 0, pm.numb, sUSrc, C.Uval, part, wave, lane, NW);
      else reduce_umulti(C.umb + pm.umb0, 0, pm.numb, C.usrc + pm.us0, C.Uval, part, wave, lane, NW);
    }
  }
  SSLAM_STAMP(6)
  // ---- 4. one coalesced stream out
  if (C.flat_L) {
    D2* dst = reinterpret_cast<D2*>(C.Lval + pm.lbase);
    const D2* src = reinterpret_cast<const D2*>(smL);
    for (int e = tid; e < (pm.lsize >> 1); e += NT) dst[e] = src[e];
  } else {
    // class-interleaved form (chol_plan.hpp): every size class of the piece transposed, element k of its i-th block at k * n + i;
    // consecutive lanes still write consecutive addresses.  t / n by a float reciprocal (exact for t < 2^22).
    double* dst = C.Lval + pm.lbase;
    const int nA = pm.n36, nB = pm.n18, nC = pm.nb - nA - nB, eA = 36 * nA, eB = eA + 18 * nB;
    const float rA = 1.0f / (float)max(nA, 1), rB = 1.0f / (float)max(nB, 1), rC = 1.0f / (float)max(nC, 1);
    for (int t = tid; t < eA; t += NT) { const int k = (int)(((float)t + 0.5f) * rA), i = t - k * nA; dst[t] = smL[i * 36 + k]; }
    for (int t = tid; t < 18 * nB; t += NT) { const int k = (int)(((float)t + 0.5f) * rB), i = t - k * nB; dst[eA + t] = smL[eA + i * 18 + k]; }
    for (int t = tid; t < 9 * nC; t += NT) { const int k = (int)(((float)t + 0.5f) * rC), i = t - k * nC; dst[eB + t] = smL[eB + i * 10 + k]; }
  }
  for (int e = tid; e < pm.ysize; e += NT) C.y[pm.y0 + e] = smY[e];
  SSLAM_STAMP(7)
  if (dbg && threadIdx.x == 0) { dbg[8] += 1; dbg[9] += pm.nilv; dbg[10] += pm.nuit; dbg[11] += pm.nit_i; }
}
#undef SSLAM_STAMP

template <int NT, bool USTAGE, bool RIGHT = false>   // RIGHT: mid pieces (several columns of a chain per piece), internal updates by source column
__global__ __launch_bounds__(NT, 4) void k_chol_pieces(BatchView V, CholView C, int begin, const int* __restrict__ idx) {   // <= 128 VGPRs: four waves per SIMD
  extern __shared__ double sm[];
  const PieceMeta pm = C.lpiece[idx ? idx[blockIdx.x] : begin + blockIdx.x];   // idx: the pieces of the graphs that are still active
  if (!V.lm[pm.graph].in_trial) return;
  chol_piece<NT, USTAGE, RIGHT>(V, C, pm, sm, (C.dbg && blockIdx.x == 0) ? C.dbg + 16 : nullptr);
}

// Top of the elimination tree: once a graph is down to a few pieces per depth a launch per depth only buys launch
// latency.  One workgroup per graph walks its remaining pieces in elimination order; the barrier between pieces orders
// the L / y stores of one piece before the loads of the next (same CU).
template <int NT>
__global__ __launch_bounds__(NT) void k_chol_tail(BatchView V, CholView C, const int* __restrict__ idx) {
  extern __shared__ double sm[];
  const int g = idx ? idx[blockIdx.x] : blockIdx.x;
  if (!V.lm[g].in_trial) return;
  const int q1 = C.tail_ptr[g + 1];
  for (int q = C.tail_ptr[g]; q < q1; ++q) {
    if (C.rupd) chol_piece<NT, false, true>(V, C, C.lpiece[C.ltail0 + q], sm, (C.dbg && g == 0) ? C.dbg : nullptr);
    else chol_piece<NT, false, false>(V, C, C.lpiece[C.ltail0 + q], sm, (C.dbg && g == 0) ? C.dbg : nullptr);
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// backward substitution of one piece:  x_j = L_jj^-T (y_j - sum_i L_ij^T x_i),  columns of the piece top-down.
// The factor is read in its class-interleaved HBM form, one thread per block, every element once, consecutive lanes on consecutive
// addresses -- no LDS copy of the piece: a block whose row lies above the piece is multiplied with its (final) x_i out of
// registers straight away; only the blocks with their row inside the piece are parked in LDS for the levels.
// LDS: [row-in-piece blocks, 36 doubles each | y -> x of the piece | contributions of the rows above, 6 per block | block table | column table]
// ------------------------------------------------------------------------------------------------
#define SSLAM_BSTAMP(k)                                                         \
  if (dbg) {                                                                    \
    const long long now_ = clock64();                                           \
    if (threadIdx.x == 0) dbg[k] += now_ - tprev;                               \
    tprev = now_;                                                               \
  }
template <int NT>
__device__ __forceinline__ void chol_piece_backward(const CholView& C, const PieceMeta pm, const double* __restrict__ y, double* x, double* sm, long long* dbg) {
  ILevel* s_lvb = reinterpret_cast<ILevel*>(sm);
  long long tprev = dbg ? clock64() : 0;
  const int tid = threadIdx.x;
  const int Yp = (pm.ysize + 1) & ~1;
  double* smI = sm + 4 * pm.nilv;
  double* smX = smI + 36 * pm.nint;
  double* smE = smX + Yp;                                   // [nb][6]
  int2* sBlk = reinterpret_cast<int2*>(smE + 6 * pm.nb);    // {LDS offset of a row-in-piece block | di << 24 | dj << 28, local y offset of the row}
  int4* sCol = reinterpret_cast<int4*>(sBlk + pm.nb + (pm.nb & 1));   // {first block (piece-local), dim | nb << 8 | nbi << 20, y offset, -}
  int* sXoff = reinterpret_cast<int*>(sCol + pm.nc);
  const int nA = pm.n36, nB = pm.n18, nC = pm.nb - nA - nB, eA = 36 * nA, eB = eA + 18 * nB;
  // ---- 0. one pass over the blocks
  const double y_r = y[pm.y0 + min(tid, pm.ysize - 1)];
  const ILevel lv_r = C.ilv[pm.ilv0 + min(tid, max(pm.nilv - 1, 0))];
  const ColMeta col_r = C.col[pm.c0 + min(tid, pm.nc - 1)];
  for (int b = tid; b < pm.nb; b += NT) {
    const BlkMeta bm = C.blk[pm.b0 + b];
    const int di = bm.info & 15, dj = (bm.info >> 4) & 15;
    const bool in = bm.info & kBlkRowIn;
    const int ol = bm.off - pm.lbase;
    int st, n, i;
    if (ol < eA) { st = 0; n = nA; i = ol / 36; }
    else if (ol < eB) { st = eA; n = nB; i = (ol - eA) / 18; }
    else { st = eB; n = nC; i = (ol - eB) / 10; }
    const double* __restrict__ g = C.Lval + pm.lbase + st + i;
    if (in) {
      const int lo = 36 * blk_irank(bm.info);
      sBlk[b] = make_int2(lo | (di << 24) | (dj << 28), bm.yoff_row - pm.y0);
      double* o = smI + lo;
      if (dj == 6) {
        for (int r = 0; r < di; r += 3) {
          double v[18];
#pragma unroll
          for (int k = 0; k < 18; ++k) v[k] = g[(size_t)(r * 6 + k) * n];
#pragma unroll
          for (int k = 0; k < 18; ++k) o[r * 6 + k] = v[k];
        }
      } else {
        for (int r = 0; r < di; r += 3) {
          double v[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) v[k] = g[(size_t)(r * 3 + k) * n];
#pragma unroll
          for (int k = 0; k < 9; ++k) o[r * 3 + k] = v[k];
        }
      }
    } else {
      const double* xi = x + bm.xoff_row;
      double sv[6] = {0, 0, 0, 0, 0, 0};
      if (dj == 6) {
        for (int r = 0; r < di; r += 3) {
          double v[18];
          const double x0 = xi[r], x1 = xi[r + 1], x2 = xi[r + 2];
#pragma unroll
          for (int k = 0; k < 18; ++k) v[k] = g[(size_t)(r * 6 + k) * n];
#pragma unroll
          for (int c = 0; c < 6; ++c) { sv[c] += v[c] * x0; sv[c] += v[6 + c] * x1; sv[c] += v[12 + c] * x2; }   // row by row, like the dense product
        }
      } else {
        for (int r = 0; r < di; r += 3) {
          double v[9];
          const double x0 = xi[r], x1 = xi[r + 1], x2 = xi[r + 2];
#pragma unroll
          for (int k = 0; k < 9; ++k) v[k] = g[(size_t)(r * 3 + k) * n];
#pragma unroll
          for (int c = 0; c < 3; ++c) { sv[c] += v[c] * x0; sv[c] += v[3 + c] * x1; sv[c] += v[6 + c] * x2; }
        }
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) smE[b * 6 + c] = sv[c];
    }
  }
  // the small tables: loaded above (in flight together with the block records), stored here
  if (tid < pm.ysize) smX[tid] = y_r;
  for (int e = tid + NT; e < pm.ysize; e += NT) smX[e] = y[pm.y0 + e];
  if (tid < pm.nilv) s_lvb[tid] = lv_r;
  for (int i = tid + NT; i < pm.nilv; i += NT) s_lvb[i] = C.ilv[pm.ilv0 + i];
  if (tid < pm.nc) { sCol[tid] = make_int4(col_r.b0 - pm.b0, col_r.dim | (col_r.nb << 8) | (col_r.nbi << 20), col_r.yoff - pm.y0, 0); sXoff[tid] = col_r.xoff; }
  for (int c = tid + NT; c < pm.nc; c += NT) {
    const ColMeta cm = C.col[pm.c0 + c];
    sCol[c] = make_int4(cm.b0 - pm.b0, cm.dim | (cm.nb << 8) | (cm.nbi << 20), cm.yoff - pm.y0, 0);
    sXoff[c] = cm.xoff;
  }
  __syncthreads();
  SSLAM_BSTAMP(0)
  // ---- 1. rows above the piece: summed per column in block order
  for (int t = tid; t < pm.nc * 8; t += NT) {
    const int cidx = t >> 3, c = t & 7;
    const int4 cm = sCol[cidx];
    const int dj = cm.y & 255, nb = (cm.y >> 8) & 0xFFF, nbi = cm.y >> 20;
    if (c >= dj) continue;
    double acc = 0;
    for (int bi = nbi; bi < nb; ++bi) acc += smE[(cm.x + bi) * 6 + c];
    smX[cm.z + c] -= acc;
  }
  __syncthreads();
  SSLAM_BSTAMP(1)
  // ---- 2. the levels inside the piece, top-down; a team of 8 Q lanes per column (lane = component c of block slice q)
  for (int il = pm.nilv - 1; il >= 0; --il) {
    const ILevel lv = s_lvb[il];
    const int ncols = lv.c1 - lv.c0;
    int Q = 1;
    while (Q < 8 && ncols * 16 * Q <= NT) Q *= 2;
    const int team = tid / (8 * Q), lt = tid % (8 * Q), nteams = NT / (8 * Q);
    const int c = lt & 7, q = lt >> 3;
    for (int ci = team; ci < ncols; ci += nteams) {
      const int4 cm = sCol[lv.c0 - pm.c0 + ci];
      const int dj = cm.y & 255, nbi = cm.y >> 20;
      const int cc = min(c, dj - 1);   // idle lanes of the team shadow the last component
      const double* D = smI + (sBlk[cm.x].x & 0xFFFFFF);
      double acc = 0;
      for (int bi = 1 + q; bi < nbi; bi += Q) {
        const int2 bm = sBlk[cm.x + bi];
        const double* Bk = smI + (bm.x & 0xFFFFFF) + cc;
        const double* xi = smX + bm.y;
        double s = Bk[0] * xi[0] + Bk[dj] * xi[1] + Bk[2 * dj] * xi[2];
        if (((bm.x >> 24) & 15) == 6) s += Bk[3 * dj] * xi[3] + Bk[4 * dj] * xi[4] + Bk[5 * dj] * xi[5];
        acc += s;
      }
      if (Q > 1) acc += __shfl_xor(acc, 8, 64);
      if (Q > 2) acc += __shfl_xor(acc, 16, 64);
      if (Q > 4) acc += __shfl_xor(acc, 32, 64);
      // every lane of the team redoes the small triangular solve  x = L_jj^-T t  on its own: one LDS hand-over of t instead of a
      // chain of six shuffles (the divisions stay: x_r = a / L_rr is what the reference's csparse back-substitution computes, and
      // LM's accept / reject decisions at convergence follow the last bit)
      if (q == 0 && c < dj) smX[cm.z + c] -= acc;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double tv[6], xv[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) tv[r] = smX[cm.z + min(r, dj - 1)];
      if (dj == 6) {
#pragma unroll
        for (int r = 5; r >= 0; --r) {
          double a = tv[r];
#pragma unroll
          for (int s2 = 5; s2 > r; --s2) a -= D[s2 * 6 + r] * xv[s2];
          xv[r] = a / D[r * 6 + r];
        }
      } else {
#pragma unroll
        for (int r = 2; r >= 0; --r) {
          double a = tv[r];
#pragma unroll
          for (int s2 = 2; s2 > r; --s2) a -= D[s2 * 3 + r] * xv[s2];
          xv[r] = a / D[r * 3 + r];
        }
        xv[3] = xv[4] = xv[5] = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      if (q == 0 && c < dj) smX[cm.z + c] = c == 0 ? xv[0] : c == 1 ? xv[1] : c == 2 ? xv[2] : c == 3 ? xv[3] : c == 4 ? xv[4] : xv[5];
    }
    __syncthreads();
  }
  SSLAM_BSTAMP(2)
  // ---- 3. x of the piece -> HBM (internal row order), one thread per column
  for (int cidx = tid; cidx < pm.nc; cidx += NT) {
    const int4 cm = sCol[cidx];
    const int dj = cm.y & 255;
    double* xo = x + sXoff[cidx];
#pragma unroll
    for (int c = 0; c < 6; ++c) if (c < dj) xo[c] = smX[cm.z + c];
  }
  SSLAM_BSTAMP(3)
  if (dbg && threadIdx.x == 0) { dbg[4] += 1; dbg[5] += pm.nilv; dbg[6] += pm.nb; }
}
#undef SSLAM_BSTAMP

template <int NT>
__global__ __launch_bounds__(NT, NT == 64 ? 8 : 1) void k_chol_back_pieces(CholView C, int begin, const double* __restrict__ y, double* x, const LmState* __restrict__ lm, const int* __restrict__ idx) {
  extern __shared__ double sm[];
  const PieceMeta pm = C.lpiece[idx ? idx[blockIdx.x] : begin + blockIdx.x];
  if (lm && !lm[pm.graph].in_trial) return;
  chol_piece_backward<NT>(C, pm, y, x, sm, (C.dbg && blockIdx.x == 0) ? C.dbg + 40 : nullptr);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_chol_back_tail(CholView C, const double* __restrict__ y, double* x, const LmState* __restrict__ lm, const int* __restrict__ idx) {
  extern __shared__ double sm[];
  const int g = idx ? idx[blockIdx.x] : blockIdx.x;
  if (lm && !lm[g].in_trial) return;
  const int q0 = C.tail_ptr[g];
  for (int q = C.tail_ptr[g + 1] - 1; q >= q0; --q) {
    chol_piece_backward<NT>(C, C.lpiece[C.ltail0 + q], y, x, sm, (C.dbg && g == 0) ? C.dbg + 32 : nullptr);
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// level-scheduled multi right-hand-side solves on the finished factor (marginals): blockIdx.y = right-hand side
// ------------------------------------------------------------------------------------------------
// forward: y_j = L_jj^-1 (b_j - sum_k L_jk y_k); rhs in internal row order, y in elimination order
__global__ __launch_bounds__(64) void k_chol_forward_level(CholView C, int lvl_begin, const double* __restrict__ rhs, double* __restrict__ y) {
  __shared__ double t[8];
  const int j = C.lvl_cols[lvl_begin + blockIdx.x];
  const ColMeta cm = C.col[j];
  const size_t vo = (size_t)blockIdx.y * C.dim;
  const int lane = threadIdx.x;
  const int dj = cm.dim;
  const double* __restrict__ L = C.Lval;
  if (lane < dj) {
    double a = rhs[vo + cm.xoff + lane];
    for (int u = cm.f0; u < cm.f1; ++u) {   // the blocks of row j, ascending k
      const FwdMeta fm = C.fwd[u];
      const int dk = fm.off < 0 ? 6 : 3;
      const double* pa = L + (fm.off & 0x7FFFFFFF) + lane * dk;
      const double* yk = y + vo + fm.yoff;
      for (int q = 0; q < dk; ++q) a -= pa[q] * yk[q];
    }
    t[lane] = a;
  }
  __syncthreads();
  if (lane == 0) {
    const double* D = L + cm.base;
    for (int r = 0; r < dj; ++r) {
      double a = t[r];
      for (int s = 0; s < r; ++s) a -= D[r * dj + s] * t[s];
      t[r] = a / D[r * dj + r];
    }
    for (int r = 0; r < dj; ++r) y[vo + cm.yoff + r] = t[r];
  }
}

// backward substitution of one column by a team of 8 * Q lanes:   x_j = L_jj^-T (y_j - sum_i L_ij^T x_i)
template <int Q>
__device__ __forceinline__ void chol_backward_column(const CholView& C, const int j, const double* __restrict__ y, double* x,
                                                     const size_t vo, const int lt) {
  const int c = lt & 7, q = lt >> 3;
  const ColMeta cm = C.col[j];
  const int dj = cm.dim;
  const double* __restrict__ L = C.Lval;
  const int cc = min(c, dj - 1);
  const double* D = L + cm.base;
  double acc = 0;
  for (int bi = 1 + q; bi < cm.nb; bi += Q) {
    const BlkMeta bm = C.blk[cm.b0 + bi];
    const double* Bk = L + bm.off + cc;
    const double* xi = x + vo + bm.xoff_row;
    double s = Bk[0] * xi[0] + Bk[dj] * xi[1] + Bk[2 * dj] * xi[2];
    if ((bm.info & 15) == 6) s += Bk[3 * dj] * xi[3] + Bk[4 * dj] * xi[4] + Bk[5 * dj] * xi[5];
    acc += s;
  }
  if (Q > 1) acc += __shfl_xor(acc, 8, 64);
  if (Q > 2) acc += __shfl_xor(acc, 16, 64);
  if (Q > 4) acc += __shfl_xor(acc, 32, 64);
  double t = y[vo + cm.yoff + cc] - acc;
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    const int rr = min(r, dj - 1);
    const double drr = D[rr * dj + rr], drc = D[rr * dj + cc];
    const double xr = __shfl(t, rr, 8) / drr;
    if (r < dj) {
      if (c == r) t = xr;
      else if (c < r) t -= drc * xr;
    }
  }
  if (q == 0 && c < dj) x[vo + cm.xoff + c] = t;
}

template <int Q>
__global__ __launch_bounds__(64) void k_chol_backward_level(CholView C, int lvl_begin, int n, const double* __restrict__ y, double* x) {
  constexpr int kCols = 8 / Q;
  const int p = blockIdx.x * kCols + threadIdx.x / (8 * Q);
  const int j = C.lvl_cols[lvl_begin + min(p, n - 1)];
  if (p < n) chol_backward_column<Q>(C, j, y, x, (size_t)blockIdx.y * C.dim, threadIdx.x % (8 * Q));
}

// ------------------------------------------------------------------------------------------------
// Diagonal blocks of H^-1 = (L L^T)^-1 for a list of vertices in ONE launch (computeLandmarkMarginals, reference
// src/ps_graph_slam/graph_slam.cpp:221-234 -> g2o MarginalCovarianceCholesky).  Z(v,v) = E_v^T L^-T L^-1 E_v = Y^T Y with Y = L^-1 E_v,
// and Y is non-zero only on the PATH from v's column to the root of the elimination tree; every off-diagonal block of a column on that
// path has its row further up the same path (struct(k) is a subset of the ancestors of k).  One wave per vertex walks its path bottom-up
// with a right-looking forward substitution held in LDS: finalise Y_s = L_ss^-1 acc_s, then every block (i, path[s]) of the column
// subtracts L(i, s) Y_s from acc_i -- one lane per block, distinct rows, no conflicts, fixed order.  Work O(path^2) per vertex instead of
// two triangular solves over the whole factor per right-hand side column (round 3: 2 x levels launches, 3 N_l right-hand sides and the
// whole solution matrix over PCIe).  Needs the flat factor (chol_factor_and_forward(b, true)).
// LDS: [Y: maxlen x 36 | diagonal block 36 | path records: {first block, blocks, diagonal offset, dim | yoff << 8} maxlen x int4]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chol_marginal_paths(CholView C, const int* __restrict__ path_ptr, const int* __restrict__ req_dim,
                                                            const int* __restrict__ path_cols, double* __restrict__ out, int maxlen) {
  extern __shared__ double sm[];
  const int r = blockIdx.x, lane = threadIdx.x;
  const int p0 = path_ptr[r], P = path_ptr[r + 1] - p0, D = req_dim[r];
  double* Y = sm;
  double* sD = Y + (size_t)maxlen * 36;
  int4* prec = reinterpret_cast<int4*>(sD + 36);
  int* pyoff = reinterpret_cast<int*>(prec + maxlen);
  const double* __restrict__ L = C.Lval;
  for (int t = lane; t < P; t += 64) {
    const ColMeta cm = C.col[path_cols[p0 + t]];
    prec[t] = make_int4(cm.b0, cm.nb, cm.base, cm.dim);
    pyoff[t] = cm.yoff;
  }
  for (int e = lane; e < P * 36; e += 64) Y[e] = 0.0;
  __syncthreads();
  if (lane < D) Y[lane * 6 + lane] = 1.0;   // E_v: identity in the vertex' own rows (path entry 0)
  __syncthreads();
  for (int s = 0; s < P; ++s) {
    const int4 pr = prec[s];
    const int da = pr.w;
    if (lane < da * da) sD[lane] = L[pr.z + lane];
    __syncthreads();
    if (lane < D) {   // one lane per right-hand side column: Y_s = L_ss^-1 acc_s
      double yv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i < da) {
          double a = Y[s * 36 + i * 6 + lane];
#pragma unroll
          for (int m = 0; m < 6; ++m) if (m < i) a -= sD[i * da + m] * yv[m];
          yv[i] = a / sD[i * da + i];
          Y[s * 36 + i * 6 + lane] = yv[i];
        }
      }
    }
    __syncthreads();
    for (int bq = lane; bq < pr.y - 1; bq += 64) {   // the blocks below the diagonal: acc_t -= L(t, s) Y_s
      const BlkMeta bm = C.blk[pr.x + 1 + bq];
      const int di = bm.info & 15;
      int lo = s + 1, hi = P - 1;                    // the row's place on the path (y offsets ascend along it)
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (pyoff[mid] < bm.yoff_row) lo = mid + 1; else hi = mid; }
      const double* Lb = L + bm.off;
      double* acc = Y + lo * 36;
      for (int i = 0; i < di; ++i) {
        double lrow[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) lrow[m] = m < da ? Lb[i * da + m] : 0.0;
        for (int c = 0; c < D; ++c) {
          double a = 0;
#pragma unroll
          for (int m = 0; m < 6; ++m) a += lrow[m] * Y[s * 36 + m * 6 + c];   // rows >= da of Y_s are zero
          acc[i * 6 + c] -= a;
        }
      }
    }
    __syncthreads();
  }
  if (lane < D * D) {
    const int rr = lane / D, cc = lane - rr * D;
    double z = 0;
    for (int t = 0; t < P; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) z += Y[t * 36 + i * 6 + rr] * Y[t * 36 + i * 6 + cc];
    out[(size_t)r * 36 + lane] = z;
  }
}

// ------------------------------------------------------------------------------------------------
// One LM iteration of a SMALL graph after its linearisation, in ONE launch (the orchestrator's tick re-optimises a
// graph of a few hundred vertices 25-40 trials per tick, and ~65 launches per trial made the MI355X slower than one host core).
// One workgroup per graph whose whole elimination tree is walked by the tail kernels (plans with no per-depth launches):
//   begin step (lambda = tau max diag on the first) -> factor (the tail pieces, chol_piece) -> backward substitution -> x [+] dx ->
//   chi2 of the trial -> gain ratio, accept / reject (lm_control_apply) -> commit,
// and, while the trial is rejected, again with the raised lambda -- until the graph needs a new linearisation (the host launches the
// Jacobian kernels and this kernel once per LM iteration) or terminates.  Every sum runs over the same chunks in the same order as the
// stand-alone kernels (k_chi2 / k_scale / k_lm_control): results are bitwise those of the unfused path
// (tests/test_graph_gpu.py::test_fused_small_graph_trials_equal_the_unfused_path).
// ------------------------------------------------------------------------------------------------
template <int BS, int NT>
__device__ __forceinline__ void vblock_store_sum(double v, double* red, double* dst, bool live) {   // block_sum<BS> of every BS-thread slice of the workgroup
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (live && tid % BS == 0) {
    double s = 0;
    for (int k = 0; k < BS / 64; ++k) s += red[(tid >> 6) + k];
    *dst = s;
  }
  __syncthreads();
}

// the three parts of an LM step around the linear solve, one workgroup per graph (shared by the begin / end kernels
// of the single-launch solve)
template <int NT>
__device__ __forceinline__ void lm_begin_small(const BatchView& V, const CholView& C, int g, double* red) {   // k_maxdiag + k_lm_begin_step + k_chol_begin
  const int tid = threadIdx.x;
  LmState& S = V.lm[g];
  const GraphSeg sg = V.seg[g];
  const int nrows = sg.nprow * 6 + sg.nlrow * 3;
  const int lin = S.lin;
  if (lin && S.iter == 0) {
    double d = 0;
    for (int e = tid; e < nrows; e += NT) {
      const RowRef R = row_ref(V, sg, e);
      d = fmax(d, fabs(R.is_pose ? V.Hpp_diag[(size_t)R.row * 36 + R.r * 7] : V.Hll_diag[(size_t)R.row * 9 + R.r * 4]));
    }
    const double m = block_max<NT>(d, red);
    if (tid == 0) { S.max_diag = m; S.lambda = 1e-5 * m; S.nu = 2.0; }
  }
  if (tid == 0) {
    if (lin) { S.q = 0; S.rho = 0; S.in_trial = 1; S.lin = 0; }
    S.accept = 0;
    C.fail[g] = 0;
  }
  __threadfence_block();
  __syncthreads();
}
template <int NT>
__device__ __forceinline__ void lm_end_small(const BatchView& V, const CholView& C, int g, double* __restrict__ part_e, int max_iters, double* red) {
  // k_chol_end + k_oplus + k_chi2 + k_scale + k_lm_control + k_commit for graph g (in a trial)
  const int tid = threadIdx.x;
  LmState& S = V.lm[g];
  const GraphSeg sg = V.seg[g];
  const int nec = edge_chunks(sg), nrc = row_chunks(sg);
  if (tid == 0) V.pcg_fail[g] = C.fail[g];
  for (int i = tid; i < sg.nprow + sg.nlrow; i += NT) oplus_row(V, i < sg.nprow ? sg.prow0 + i : V.nPr + sg.lrow0 + (i - sg.nprow), V.x);
  __threadfence_block();
  __syncthreads();
  for (int c0 = 0; c0 < nec; c0 += NT / kEdgeChunk) {   // the chunks of k_chi2
    const int chunk = c0 + tid / kEdgeChunk;
    const double c = chunk < nec ? edge_chi2(V, sg, chunk * kEdgeChunk + tid % kEdgeChunk, V.pose_trial, V.lmk_trial) : 0.0;
    vblock_store_sum<kEdgeChunk, NT>(c, red, part_e + (size_t)g * V.maxEdgeChunks + chunk, chunk < nec);
  }
  {   // dx . (lambda dx + b): the chunks of k_scale
    const double lambda = S.lambda;
    constexpr int NV = NT / kRowChunk;
    for (int c0 = 0; c0 < nrc; c0 += NV) {
      const int vb = tid / kRowChunk, chunk = c0 + vb;
      const bool live = vb < NV && chunk < nrc;
      double v = 0;
      if (live) {
        const RowRef R = row_ref(V, sg, chunk * kRowChunk + tid % kRowChunk);
        if (R.valid) { const double d = V.x[R.xoff]; v = d * (lambda * d + V.bvec[R.xoff]); }
      }
      vblock_store_sum<kRowChunk, NT>(v, red, V.part_a + (size_t)g * V.maxRowChunks + chunk, live);
    }
  }
  __threadfence_block();
  __syncthreads();
  if (tid < 64) {   // accept / reject (k_lm_control)
    const double tchi = wave_sum_partials(part_e + (size_t)g * V.maxEdgeChunks, nec);
    const double sc = wave_sum_partials(V.part_a + (size_t)g * V.maxRowChunks, nrc);
    if (tid == 0) lm_control_apply(S, tchi, sc, V.pcg_fail[g], max_iters);
  }
  __threadfence_block();
  __syncthreads();
  if (S.accept) {   // k_commit
    for (int i = tid; i < sg.nprow + sg.nlrow; i += NT) commit_row(V, i < sg.nprow ? sg.prow0 + i : V.nPr + sg.lrow0 + (i - sg.nprow));
    __threadfence_block();
  }
  __syncthreads();
}

// the same two halves as kernels of their own, for the single-launch solve (k_chol_flow) between them: 3 launches per damping trial after the
// Jacobian kernels instead of ~12 small ones + two per depth of the tree
template <int NT>
__global__ __launch_bounds__(NT) void k_lm_begin_small(BatchView V, CholView C) {
  __shared__ double red[NT / 64];
  if (!V.lm[blockIdx.x].active) return;
  lm_begin_small<NT>(V, C, blockIdx.x, red);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_lm_end_small(BatchView V, CholView C, double* __restrict__ part_e, int max_iters) {
  __shared__ double red[NT / 64];
  if (!V.lm[blockIdx.x].active || !V.lm[blockIdx.x].in_trial) return;
  lm_end_small<NT>(V, C, blockIdx.x, part_e, max_iters, red);
}

// ---- speculative damping trials (SpecLanes): begin / per-lane end / replay of the accept-reject sequence + commit ----------------
// one thread: lane k tries the lambda the sequential loop would reach after k rejected trials; returns the number of lanes at work
__device__ __forceinline__ int spec_setup_lanes(const BatchView& V, const SpecLanes& SL, int g) {
  const LmState S = V.lm[g];
  double lam = S.lambda, nu = S.nu;
  int n = 0;
  for (int k = 0; k < SL.K; ++k) {
    LmState Lk = S;
    Lk.lambda = lam; Lk.in_trial = (S.active && S.in_trial && S.q + k < 10 && (k == 0 || S.q >= SL.after)) ? 1 : 0;
    SL.lm[(size_t)k * V.B + g] = Lk;
    SL.fail[(size_t)k * V.B + g] = 0;
    n += Lk.in_trial;
    lam *= nu; nu *= 2;
  }
  return n;
}
// x [+] dx, chi2 and dx . (lambda dx + b) of lane k (one workgroup)
template <int NT>
__device__ __forceinline__ void spec_lane_end(BatchView V, const SpecLanes& SL, const long long k, const int g, double* red) {
  const int tid = threadIdx.x;
  V.lm = SL.lm + k * V.B; V.x = SL.x + k * SL.sx; V.pose_trial = SL.pose_trial + k * SL.spose; V.lmk_trial = SL.lmk_trial + k * SL.slmk;
  V.part_a = SL.part_a + k * SL.spa;
  double* part_e = SL.part_e + k * SL.spe;
  const LmState& S = V.lm[g];
  if (!S.active || !S.in_trial) return;
  const GraphSeg sg = V.seg[g];
  const int nec = edge_chunks(sg), nrc = row_chunks(sg);
  for (int i = tid; i < sg.nprow + sg.nlrow; i += NT) oplus_row(V, i < sg.nprow ? sg.prow0 + i : V.nPr + sg.lrow0 + (i - sg.nprow), V.x);
  // vertices without a row (fixed: the gauge vertex) keep their estimate in the lane's trial arrays too
  const double* __restrict__ cur_pose = SL.pose_cur;
  const double* __restrict__ cur_lmk = SL.lmk_cur;
  for (int i = tid; i < sg.npose; i += NT) if (V.pose_row[sg.pose0 + i] < 0) for (int q = 0; q < 8; ++q) V.pose_trial[(size_t)(sg.pose0 + i) * 8 + q] = cur_pose[(size_t)(sg.pose0 + i) * 8 + q];
  for (int i = tid; i < sg.nlm; i += NT) if (V.lm_row[sg.lm0 + i] < 0) for (int q = 0; q < 4; ++q) V.lmk_trial[(size_t)(sg.lm0 + i) * 4 + q] = cur_lmk[(size_t)(sg.lm0 + i) * 4 + q];
  __threadfence_block();
  __syncthreads();
  for (int c0 = 0; c0 < nec; c0 += NT / kEdgeChunk) {
    const int chunk = c0 + tid / kEdgeChunk;
    const double c = chunk < nec ? edge_chi2(V, sg, chunk * kEdgeChunk + tid % kEdgeChunk, V.pose_trial, V.lmk_trial) : 0.0;
    vblock_store_sum<kEdgeChunk, NT>(c, red, part_e + (size_t)g * V.maxEdgeChunks + chunk, chunk < nec);
  }
  const double lambda = S.lambda;
  constexpr int NV = NT / kRowChunk;
  for (int c0 = 0; c0 < nrc; c0 += NV) {
    const int vb = tid / kRowChunk, chunk = c0 + vb;
    const bool live = vb < NV && chunk < nrc;
    double v = 0;
    if (live) {
      const RowRef R = row_ref(V, sg, chunk * kRowChunk + tid % kRowChunk);
      if (R.valid) { const double d = V.x[R.xoff]; v = d * (lambda * d + V.bvec[R.xoff]); }
    }
    vblock_store_sum<kRowChunk, NT>(v, red, V.part_a + (size_t)g * V.maxRowChunks + chunk, live);
  }
}
// OptimizationAlgorithmLevenberg's do { ... } while (rho < 0 && q < 10) over the finished lanes, in order, and the commit of the accepted one
template <int NT>
__device__ __forceinline__ void spec_control(const BatchView& V, const SpecLanes& SL, const int g, const int max_iters, int* s_lane) {
  const int tid = threadIdx.x;
  LmState& S = V.lm[g];
  if (!S.active || !S.in_trial) return;
  const GraphSeg sg = V.seg[g];
  const int nec = edge_chunks(sg), nrc = row_chunks(sg);
  if (tid == 0) *s_lane = -1;
  __syncthreads();
  if (tid < 64) {
    for (int k = 0; k < SL.K; ++k) {
      if (!SL.lm[(size_t)k * V.B + g].in_trial) break;
      const double tchi = wave_sum_partials(SL.part_e + (size_t)k * SL.spe + (size_t)g * V.maxEdgeChunks, nec);
      const double sc = wave_sum_partials(SL.part_a + (size_t)k * SL.spa + (size_t)g * V.maxRowChunks, nrc);
      int more = 0;
      if (tid == 0) {
        const int failed = SL.fail[(size_t)k * V.B + g];
        V.pcg_fail[g] = failed;
        lm_control_apply(S, tchi, sc, failed, max_iters);
        if (S.accept) *s_lane = k;
        more = S.in_trial;
      }
      more = __shfl(more, 0, 64);
      if (!more) break;
    }
    // (trials left and no lane left -- the adaptive form after the first rejected trial of an iteration: S carries lambda, nu and q on, the
    // next round's lanes continue the sequence)
  }
  __threadfence_block();
  __syncthreads();
  const int lane = *s_lane;
  if (lane >= 0) {   // commit the accepted lane's estimates
    const double* pt = SL.pose_trial + (size_t)lane * SL.spose;
    const double* lt = SL.lmk_trial + (size_t)lane * SL.slmk;
    for (int i = tid; i < sg.nprow + sg.nlrow; i += NT) {
      if (i < sg.nprow) { const int pi = V.prow_pose[sg.prow0 + i]; for (int q = 0; q < 7; ++q) V.pose[(size_t)pi * 8 + q] = pt[(size_t)pi * 8 + q]; }
      else { const int li = V.lrow_lm[sg.lrow0 + (i - sg.nprow)]; for (int q = 0; q < 4; ++q) V.lmk[(size_t)li * 4 + q] = lt[(size_t)li * 4 + q]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Factorisation + both triangular solves of a SMALL batch in ONE launch, driven by the dependencies of the piece tree instead of a launch
// per depth (~26 + 26 launches per damping trial; a graph of the orchestrator's size spends its time between launches, and a
// single 5000-pose graph is latency-bound the same way).  A persistent grid of workgroups walks the pieces in launch order (children
// before parents); workgroup w takes pieces w, w + G, w + 2G, ...  A piece waits until all its child pieces have signalled (a counter per
// piece, release / acquire at agent scope: the children's update matrices cross CUs and XCDs), is factored by the same chol_piece() the
// per-depth kernels run, and signals its parent.  The backward substitution follows in the same launch, top-down: a piece waits for its
// parent's x.  The grid never exceeds what the device holds at once (a waiting workgroup must not keep the one it waits for off the chip);
// counters are never reset -- launch number `epoch` waits for epoch x (children).  A wait that does not end (it cannot, short of a lost
// workgroup) gives up after ~2^22 polls and raises the error flag instead of hanging the GPU.
// ------------------------------------------------------------------------------------------------
// fail (optional): the failure flag of the graph the waiting piece belongs to -- a wait that gives up must not let the trial go on as if the
// data it waited for were there: the graph's trial counts as a failed factorisation (lm_control rejects it) and the host reports the flag
__device__ __forceinline__ bool flow_wait(const int* p, int target, int* err, int* fail) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1 << 22)) { atomicExch(err, 1); if (fail) atomicExch(fail, 1); return false; }
  }
  return true;
}
}  // namespace sslam
#include "front_kernels.hpp"
namespace sslam {

template <int NT, bool USTAGE>
__global__ __launch_bounds__(NT) void k_chol_flow(BatchView V, CholView C, int q_first, int np, int epoch, const int2* __restrict__ dep, int* flow, SpecLanes SL,
                                                  int do_backward, double* __restrict__ part_e, int max_iters, int lm_epoch) {
  extern __shared__ double sm[];
  __shared__ double red[NT / 64];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const bool defer = do_backward & 2;   // bit 1: tables and H before the wait for the children (chol_piece DEFER)
  // bit 2: a whole damping trial in this launch -- what k_lm_begin_small does before the factorisation (workgroup g for graph g, everybody
  // waits for it: lambda and in_trial are read by every piece) and what k_lm_end_small does after the backward substitution (the
  // workgroup that finishes last, by a ticket).  Two launches less per trial of the orchestrator's graphs (measured: 0.02 - 0.15 ms per
  // tick -- a trial waits for the chain of pieces inside this launch, not for launches).
  const bool lmstep = (do_backward & 4) && SL.K == 0;
  int wg = blockIdx.x, nwg = gridDim.x;
  if (SL.K > 0) {   // speculative damping trials: a lane = a range of workgroups, with its own factor, vectors, counters and lambda
    long long k = 0;
    if (wg < SL.g0) nwg = SL.g0;
    else { const int r = wg - SL.g0; k = 1 + r / SL.g1; wg = r - (int)(k - 1) * SL.g1; nwg = SL.g1; }
    V.lm = SL.lm + k * V.B; V.x = SL.x + k * SL.sx;
    C.Lval = SL.Lval + k * SL.sL; C.Uval = SL.Uval + k * SL.sU; C.y = SL.y + k * SL.sy; C.fail = SL.fail + k * V.B;
    flow = SL.flow + k * SL.sflow;
  }
  int* child_done = flow;
  int* back_done = flow + np;
  int* fwd_done = flow + 2 * np;
  int* err = SL.K > 0 ? SL.err : flow + 3 * np;
  int* begin_done = flow + 3 * np + 1;
  int* end_ticket = flow + 3 * np + 2;
  if (lmstep) {
    if (wg < V.B) {
      if (V.lm[wg].active) lm_begin_small<NT>(V, C, wg, red);
      __syncthreads();
      if (tid == 0) { __threadfence(); __hip_atomic_fetch_add(begin_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if (tid == 0) flow_wait(begin_done, V.B * lm_epoch, err);   // (its own count: solves without the LM halves share the counters of the pieces)
    __syncthreads();
  }
  // ---- (H + lambda I) = L L^T and y = L^-1 b, leaves to roots
  // (pieces before q_first -- the wide bottom of a large graph's tree -- were factored by per-depth launches before this one; dep[].y counts
  // the children inside [q_first, np) only)
  for (int q = q_first + wg; q < np; q += nwg) {
    const PieceMeta pm = C.lpiece[q];
    const int2 d = dep[q];
    const bool run = V.lm[pm.graph].in_trial;   // (set by the begin kernel of the step: not touched inside this launch)
    if (d.y > 0 && !(defer && run)) {
      if (tid == 0) flow_wait(child_done + q, d.y * epoch, err, C.fail + pm.graph);
      __syncthreads();
    }
    if (run && C.fblob) {   // front tables (front_kernels.hpp): tables before the wait whenever the piece waits
      const int* wp = (d.y > 0 && defer) ? child_done + q : nullptr;
      if (q >= C.ltail0) front_piece<NT, true>(V, C, pm, C.lfgrp[q], sm, nullptr, wp, d.y * epoch, err);
      else front_piece<NT, false>(V, C, pm, C.lfgrp[q], sm, nullptr, wp, d.y * epoch, err);
    } else if (run) {
      const int* wp = d.y > 0 ? child_done + q : nullptr;
      if (defer) {
        if (q >= C.ltail0) {
          if (C.rupd) chol_piece<NT, false, true, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
          else chol_piece<NT, false, false, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
        } else chol_piece<NT, USTAGE, false, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
      } else if (q >= C.ltail0) {
        if (C.rupd) chol_piece<NT, false, true>(V, C, pm, sm, nullptr);
        else chol_piece<NT, false, false>(V, C, pm, sm, nullptr);
      } else chol_piece<NT, USTAGE, false>(V, C, pm, sm, nullptr);
    }
    __syncthreads();   // every thread's stores of L, y and the update matrix are issued ...
    if (tid == 0) {    // ... and released to the other CUs together with the signal
      __threadfence();
      if (d.x >= 0) __hip_atomic_fetch_add(child_done + d.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(fwd_done + q, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!(do_backward & 1)) return;   // the flat factor of the marginals: L only
  // ---- x = L^-T y, roots to leaves
  for (int i = wg; i < np - q_first; i += nwg) {
    const int q = np - 1 - i;
    const PieceMeta pm = C.lpiece[q];
    const int2 d = dep[q];
    if (tid == 0) {
      if (d.x >= 0) flow_wait(back_done + d.x, epoch, err, C.fail + pm.graph);
      else flow_wait(fwd_done + q, epoch, err, C.fail + pm.graph);
    }
    __syncthreads();
    if (V.lm[pm.graph].in_trial) chol_piece_backward<NT>(C, pm, C.y, V.x, sm, nullptr);
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(back_done + q, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (lmstep) {   // the rest of the trial: x [+] dx, chi2, accept / reject, commit -- by the workgroup that finishes last
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      const int t = __hip_atomic_fetch_add(end_ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      s_last = ((t + 1) % nwg == 0) ? 1 : 0;   // every such launch adds nwg tickets: no reset, no epoch
    }
    __syncthreads();
    if (s_last) {
      for (int g = 0; g < V.B; ++g)
        if (V.lm[g].active && V.lm[g].in_trial) lm_end_small<NT>(V, C, g, part_e, max_iters, red);
    }
  }
}

// One LM ITERATION'S WORTH of damping trials of a single small graph in ONE launch (speculative_trials): what k_lm_begin_spec, k_chol_flow over
// the lanes, k_lm_end_spec and k_lm_control_spec did in four.  Workgroup 0 begins the step and sets the lanes up (lane k: the lambda after k
// rejected trials; the adaptive form puts lanes 1.. to work only once a trial of the iteration has been rejected); everybody waits for
// that.  The workgroups of a lane that is not at work leave at once -- a lane counts its own rounds (ctl[8 + K + k]) and waits on its own
// counters with that number, so rounds it sits out cost it nothing.  A lane's workgroups factor and solve as in k_chol_flow (fetch before
// the wait); the one that takes the lane's last ticket forms the lane's chi2 / scale partial sums; the lane that finishes last replays
// g2o's accept / reject sequence over the lanes in order and commits the accepted one.  Same sums, same order: bitwise the sequential loop.
template <int NT, bool USTAGE>
__global__ __launch_bounds__(NT) void k_chol_spec_round(BatchView V, CholView C, int np, const int2* __restrict__ dep, SpecLanes SL, int round, int max_iters) {
  extern __shared__ double sm[];
  __shared__ double red[NT / 64];
  __shared__ int s_flag, s_lane;
  const int tid = threadIdx.x;
  const BatchView V0 = V;   // the graph's own state and estimates
  const int g = 0;          // (the lanes exist for batches of one graph)
  int* ctl = SL.ctl;
  if (blockIdx.x == 0) {
    if (V0.lm[g].active) lm_begin_small<NT>(V0, C, g, red);
    __syncthreads();
    if (tid == 0) {
      ctl[1] = spec_setup_lanes(V0, SL, g);
      ctl[2] = 0;
      __threadfence();
      __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) flow_wait(ctl, round, SL.err);
  __syncthreads();
  int wg = blockIdx.x, nwg = SL.g0;
  long long k = 0;
  if (wg >= SL.g0) { const int r = wg - SL.g0; k = 1 + r / SL.g1; wg = r - (int)(k - 1) * SL.g1; nwg = SL.g1; }
  V.lm = SL.lm + k * V.B; V.x = SL.x + k * SL.sx;
  C.Lval = SL.Lval + k * SL.sL; C.Uval = SL.Uval + k * SL.sU; C.y = SL.y + k * SL.sy; C.fail = SL.fail + k * V.B;
  if (!V.lm[g].in_trial) return;   // this lane sits the round out (uniform over the lane)
  int* flow = SL.flow + k * SL.sflow;
  int* child_done = flow;
  int* back_done = flow + np;
  int* fwd_done = flow + 2 * np;
  int* err = SL.err;
  const int epoch = ctl[8 + SL.K + k] + 1;
  for (int q = wg; q < np; q += nwg) {
    const PieceMeta pm = C.lpiece[q];
    const int2 d = dep[q];
    const int* wp = d.y > 0 ? child_done + q : nullptr;
    if (C.fblob) {
      if (q >= C.ltail0) front_piece<NT, true>(V, C, pm, C.lfgrp[q], sm, nullptr, wp, d.y * epoch, err);
      else front_piece<NT, false>(V, C, pm, C.lfgrp[q], sm, nullptr, wp, d.y * epoch, err);
    } else if (q >= C.ltail0) {
      if (C.rupd) chol_piece<NT, false, true, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
      else chol_piece<NT, false, false, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
    } else chol_piece<NT, USTAGE, false, true>(V, C, pm, sm, nullptr, wp, d.y * epoch, err);
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      if (d.x >= 0) __hip_atomic_fetch_add(child_done + d.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(fwd_done + q, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int i = wg; i < np; i += nwg) {
    const int q = np - 1 - i;
    const PieceMeta pm = C.lpiece[q];
    const int2 d = dep[q];
    if (tid == 0) {
      if (d.x >= 0) flow_wait(back_done + d.x, epoch, err, C.fail + pm.graph);
      else flow_wait(fwd_done + q, epoch, err, C.fail + pm.graph);
    }
    __syncthreads();
    chol_piece_backward<NT>(C, pm, C.y, V.x, sm, nullptr);
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(back_done + q, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- the lane's last workgroup: its partial sums
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const int t = __hip_atomic_fetch_add(ctl + 8 + k, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_flag = ((t + 1) % nwg == 0) ? 1 : 0;   // a lane at work adds nwg tickets per round, a lane that sits out none
  }
  __syncthreads();
  if (!s_flag) return;
  spec_lane_end<NT>(V0, SL, k, g, red);
  __syncthreads();
  if (tid == 0) {
    ctl[8 + SL.K + k] = epoch;
    __threadfence();
    const int done = __hip_atomic_fetch_add(ctl + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    s_flag = (done == ctl[1]) ? 1 : 0;
  }
  __syncthreads();
  if (!s_flag) return;
  // ---- the round's last lane: accept / reject in g2o's order, commit
  spec_control<NT>(V0, SL, g, max_iters, &s_lane);
}

__global__ void k_chol_begin(BatchView V, CholView C) {  // clear failure flags of the graphs being solved
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) C.fail[g] = 0;
}
__global__ void k_chol_end(BatchView V, CholView C) {  // publish failures through the solver-agnostic flag
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) V.pcg_fail[g] = C.fail[g];
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
namespace {
template <typename T>
int up_to_dev(CholPlan& P, hipStream_t s, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  const size_t n = h.size() + 4;   // slack: the kernels' clamped table loads may read one record past an empty range
  if (P.arena) {
    p = P.arena->take(n * sizeof(T));
    if (!p) return set_error(SSLAM_ERR_HIP, "device allocation of %zu bytes failed", n * sizeof(T));
    if (char* m = P.arena->shadow(p, n * sizeof(T))) {   // travels with the arena's next flush (one copy for all tables)
      memset(m, 0, n * sizeof(T));
      if (!h.empty()) memcpy(m, h.data(), h.size() * sizeof(T));
      *out = (const T*)p;
      return 0;
    }
    P.arena->note_direct(p, n * sizeof(T));
  } else { SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T))); P.allocs.push_back(p); }
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), s));
  if (!h.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  *out = (const T*)p;
  return 0;
}
}  // namespace

void chol_sym_input(const Batch& b, SymIn& in) {
  in.B = b.V.B; in.nPr = b.V.nPr; in.nLr = b.V.nLr;
  in.seg.resize(b.seg.size());
  for (size_t g = 0; g < b.seg.size(); ++g) in.seg[g] = SymGraph{b.seg[g].prow0, b.seg[g].nprow, b.seg[g].lrow0, b.seg[g].nlrow};
  in.ppoff = b.ppoff; in.plblk = b.plblk; in.llblk = b.llblk;
  in.hll_base = b.hll_base; in.hpp_off_base = b.hpp_off_base; in.hpl_base = b.hpl_base; in.hll_off_base = b.hll_off_base;
}

// speculative damping trials of a single small graph: the graph's option, or SSLAM_LM_SPEC = 0 / 1 / 2 for every graph of the process
int chol_spec_mode(const Batch& b) {
  static const int env = [] { const char* e = getenv("SSLAM_LM_SPEC"); return e ? std::max(0, std::min(2, atoi(e))) : -1; }();
  return env >= 0 ? env : b.graphs[0]->opt.speculative;
}
int chol_plan_build(Batch& b) {
  struct HostTimer { Batch& b; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                     ~HostTimer() { b.plan_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } host_timer{b};
  if (b.chol) { chol_plan_free(b.chol); b.chol = nullptr; }
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  SymIn in;
  chol_sym_input(b, in);
  CholOpts opt;
  opt.from_env();
  // small batches (latency-bound: the orchestrator's graph, a single large graph): the dependency-driven single launch (k_chol_flow) runs
  // every piece with the tail's workgroup size (chol_opts_normalise, shared with the plan introspection of the CPU tests)
  const int flow_mode = opt.flow;   // 0 off, 1 auto, 2 also on wide trees (SSLAM_CHOL_OPTS flow=...; read per plan: tests toggle it)
  const bool want_flow = chol_opts_normalise(opt, b.V.B, b.V.nPr + b.V.nLr);
  CholHost H;
  if (chol_symbolic(in, opt, H)) return set_error(SSLAM_ERR_NUMERIC, "Cholesky plan: %s", H.error.c_str());
  CholPlan* P = new CholPlan();
  b.chol = P;
  P->arena = b.arena;
  CholView& C = P->C;
  C.ncol = H.ncol; C.nlevels = H.nlevels; C.dim = H.dim; C.npiece = H.npiece;
  P->lvl_ptr = H.lvl_ptr; P->plv_ptr = H.plv_ptr; P->plv_lds_f = H.plv_lds_f; P->plv_lds_b = H.plv_lds_b; P->plv_nt = H.plv_nt; P->plv_cls = H.plv_cls;
  P->tail_lds_f = H.tail_lds_f; P->tail_lds_b = H.tail_lds_b; P->tail_total = (int)H.tail_pieces.size(); P->nt_tail = H.nt_tail; P->nt_ftail = H.nt_ftail; P->nt_bleaf = H.nt_bleaf; P->nt_bmid = H.nt_bmid; P->nt_btail = H.nt_btail; P->nt_leaf = H.nt_leaf; P->ustage = H.ustage;
  P->lnz = H.lnz;
  {   // elimination-tree parents (first block below the diagonal) and the vertex -> column map, for the path marginals
    std::vector<int> yoff_col(H.dim + 1, -1);
    for (int j = 0; j < H.ncol; ++j) yoff_col[H.col[j].yoff] = j;
    P->h_cparent.assign(H.ncol, -1);
    P->h_xoff_col.assign(H.dim + 1, -1);
    for (int j = 0; j < H.ncol; ++j) {
      if (H.col[j].nb > 1) P->h_cparent[j] = yoff_col[H.blk[H.col[j].b0 + 1].yoff_row];
      P->h_xoff_col[H.col[j].xoff] = j;
    }
  }
  int rc;
  if ((rc = up_to_dev(*P, b.stream, H.col, &C.col))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.blk, &C.blk))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.upd, &C.upd))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.item, &C.item))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.mb, &C.mb))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.ilv, &C.ilv))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.piece, &C.piece))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.lpiece, &C.lpiece))) return rc;
  C.ltail0 = (int)H.plv_pieces.size();
  P->lp_graph.resize(H.lpiece.size());
  for (size_t q = 0; q < H.lpiece.size(); ++q) P->lp_graph[q] = H.lpiece[q].graph;
  std::vector<int2> dep;
  if (want_flow) {
    const int np = (int)H.lpiece.size();
    std::vector<int> lidx(H.npiece, -1);
    int q = 0;
    for (int p : H.plv_pieces) lidx[p] = q++;
    for (int p : H.tail_pieces) lidx[p] = q++;
    dep.assign(np, make_int2(-1, 0));
    bool ok = q == np;
    for (int i = 0; i < np && ok; ++i) {
      const int par = H.lpiece[i].pad4;
      if (par == -2) ok = false;
      else if (par >= 0) { const int j = lidx[par]; if (j <= i) ok = false; else { dep[i].x = j; dep[j].y++; } }
    }
    P->flow = ok;
  }
  if ((rc = up_to_dev(*P, b.stream, H.asrc, &C.asrc))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.usrc, &C.usrc))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.uitem, &C.uitem))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.umb, &C.umb))) return rc;
  {   // right-looking internal updates of the tail and mid pieces (the plan drops the lists when a piece does not fit their packed records)
    C.rcol = nullptr; C.rupd = nullptr;
    if (!H.rupd.empty()) {
      if ((rc = up_to_dev(*P, b.stream, H.rcol, &C.rcol))) return rc;
      if ((rc = up_to_dev(*P, b.stream, H.rupd, &C.rupd))) return rc;
    }
  }
  if ((rc = up_to_dev(*P, b.stream, H.fwd, &C.fwd))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.lvl_cols, &C.lvl_cols))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.plv_pieces, &C.plv_pieces))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.tail_ptr, &C.tail_ptr))) return rc;
  if ((rc = up_to_dev(*P, b.stream, H.tail_pieces, &C.tail_pieces))) return rc;
  C.fblob = nullptr; C.lfgrp = nullptr;
  if (H.front) {
    if ((rc = up_to_dev(*P, b.stream, H.fblob, (const uint32_t**)&C.fblob))) return rc;
    if ((rc = up_to_dev(*P, b.stream, H.lfgrp, &C.lfgrp))) return rc;
    P->front = true; P->plv_lds_ff = H.plv_lds_ff; P->tail_lds_ff = H.tail_lds_ff;
  } else if (opt.front != 0 && chol_throughput_regime(b.V.B, b.V.nPr + b.V.nLr)) {
    fprintf(stderr, "[sslam] Cholesky plan without front tables (%s): the record kernels run\n", H.front_why.c_str());
  }
  void* p = nullptr;
  auto plan_alloc = [&](void** q, size_t bytes) -> int {
    if (P->arena) {
      *q = P->arena->take(bytes, true);
      if (*q) P->arena->note_direct(*q, bytes);   // zero-filled below with its own memset: the arena's flush must leave it alone
      return *q ? 0 : set_error(SSLAM_ERR_HIP, "device allocation of %zu bytes failed", bytes);
    }
    SSLAM_HIP_TRY(hipMalloc(q, bytes)); P->allocs.push_back(*q);
    return 0;
  };
  if ((rc = plan_alloc(&p, (H.lnz + 64) * sizeof(double)))) return rc;
  C.Lval = (double*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (H.lnz + 64) * sizeof(double), b.stream));
  const int64_t unz_all = std::max(H.unz, H.funz);   // (the front layout of the update matrices takes the same space: dense triangles either way)
  if ((rc = plan_alloc(&p, (unz_all + 64) * sizeof(double)))) return rc;
  C.Uval = (double*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (unz_all + 64) * sizeof(double), b.stream));
  P->unz = unz_all;
  if ((rc = plan_alloc(&p, (C.dim + 8) * sizeof(double)))) return rc;
  C.y = (double*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (C.dim + 8) * sizeof(double), b.stream));
  if ((rc = plan_alloc(&p, std::max(b.V.B, 1) * sizeof(int)))) return rc;
  C.fail = (int*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(C.fail, 0, std::max(b.V.B, 1) * sizeof(int), b.stream));
  C.dbg = nullptr;
  C.flat_L = 0;
  if (getenv("SSLAM_CHOL_STAMPS")) {
    if ((rc = plan_alloc(&p, 64 * sizeof(long long)))) return rc;
    C.dbg = (long long*)p;
    SSLAM_HIP_TRY(hipMemsetAsync(p, 0, 64 * sizeof(long long), b.stream));
  }
  // index lists of the LM endgame (chol_set_active), sized once: no allocation inside an optimize call (a stream group runs several of
  // them side by side)
  if (b.V.B >= 8 && !P->d_idx) {
    const size_t cap = P->lp_graph.size() + (size_t)b.V.B + 1024;
    SSLAM_HIP_TRY(hipMalloc((void**)&P->d_idx, cap * sizeof(int)));
    P->idx_cap = cap;
  }
  // LDS opt-in above 64 KiB
  size_t lds_max = (size_t)std::max(std::max(P->tail_lds_f, P->tail_lds_b), P->tail_lds_ff);
  for (size_t l = 0; l < P->plv_lds_f.size(); ++l) lds_max = std::max(lds_max, (size_t)std::max(P->plv_lds_f[l], P->plv_lds_b[l]));
  for (int v : P->plv_lds_ff) lds_max = std::max(lds_max, (size_t)v);
  lds_max *= sizeof(double);
  const int lds_lim = lds_optin_limit(b.device, 160 * 1024 - 2048, 2048);
  if (lds_max > (size_t)lds_lim) return set_error(SSLAM_ERR_UNSUPPORTED, "a piece of the factor needs %zu B of LDS (device limit %d B)", lds_max, lds_lim);
  if (lds_max > 48 * 1024) {
    // One value for every plan, set once per device under a lock: the attribute belongs to the kernel, not to a plan, and the parts of a
    // stream group build their plans side by side (a smaller value written last would fail another part's launches).
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), b.device) == done.end()) {
      const int v = lds_lim;
      const void* fns[] = {(const void*)k_chol_pieces<64, true>, (const void*)k_chol_pieces<128, true>, (const void*)k_chol_pieces<256, true>,
                           (const void*)k_chol_pieces<512, true>, (const void*)k_chol_pieces<1024, true>,
                           (const void*)k_chol_pieces<64, false>, (const void*)k_chol_pieces<128, false>, (const void*)k_chol_pieces<256, false>,
                           (const void*)k_chol_pieces<512, false>, (const void*)k_chol_pieces<1024, false>,
                           (const void*)k_chol_pieces<128, false, true>, (const void*)k_chol_pieces<256, false, true>, (const void*)k_chol_pieces<512, false, true>,
                           (const void*)k_chol_tail<512>, (const void*)k_chol_tail<1024>,
                           (const void*)k_chol_back_pieces<64>, (const void*)k_chol_back_pieces<128>, (const void*)k_chol_back_pieces<256>,
                           (const void*)k_chol_back_pieces<512>, (const void*)k_chol_back_pieces<1024>, (const void*)k_chol_back_tail<128>, (const void*)k_chol_back_tail<256>, (const void*)k_chol_back_tail<512>,
                           (const void*)k_chol_flow<512, true>, (const void*)k_chol_flow<512, false>,
                           (const void*)k_chol_spec_round<512, true>, (const void*)k_chol_spec_round<512, false>,
                           (const void*)k_front_pieces<64, false>, (const void*)k_front_pieces<128, false>, (const void*)k_front_pieces<256, false>,
                           (const void*)k_front_pieces<512, false>, (const void*)k_front_pieces<1024, false>,
                           (const void*)k_front_pieces<64, true>, (const void*)k_front_pieces<128, true>, (const void*)k_front_pieces<256, true>,
                           (const void*)k_front_pieces<512, true>, (const void*)k_front_pieces<1024, true>,
                           (const void*)k_front_tail<128>, (const void*)k_front_tail<256>, (const void*)k_front_tail<512>, (const void*)k_front_tail<1024>};
      for (const void* f : fns) SSLAM_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, v));
      done.push_back(b.device);
    }
  }
  if (P->flow) {
    // persistent grid: at most half of what the device holds at once of these workgroups (two such launches may run side by side),
    // never more workgroups than pieces
    int per_cu = 0, cus = 0;
    const void* fn = P->ustage ? (const void*)k_chol_flow<512, true> : (const void*)k_chol_flow<512, false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds_max) != hipSuccess || per_cu < 1 ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b.device) != hipSuccess || cus < 1) P->flow = false;
    else {
      // the wide bottom of a large tree keeps its per-depth launches (hundreds of independent pieces fill the chip at once; a persistent
      // grid would walk them in rounds): the single launch starts at the first depth that is no wider than its grid
      const int cap = std::max(1, per_cu * cus / 2);
      int l0 = 0;
      while (l0 < (int)P->plv_lds_f.size() && P->plv_ptr[l0 + 1] - P->plv_ptr[l0] > cap) ++l0;
      P->flow_launch0 = l0;
      // measured on one 5000-pose graph (733 / 358 / 148 / ... pieces per depth): launches for the wide depths + the single launch for the
      // rest 1.56 ms per LM iteration, a launch per depth 1.01 ms (both with 512-thread pieces; 1.50 ms with round 3's 64-thread pieces) --
      // a persistent grid walks a wide tree in rounds and pays an agent-scope release / acquire per piece.  The single launch is for trees
      // that are narrower than the grid at every depth (the orchestrator's graphs: 5.1 vs 5.7 ms per tick at 110 keyframes)
      if (l0 > 0 && flow_mode != 2) P->flow = false;
      P->flow_first = l0 < (int)P->plv_ptr.size() ? P->plv_ptr[l0] : 0;
      P->flow_grid = std::max(1, std::min((int)dep.size() - P->flow_first, cap));
      P->flow_need = std::min(1.0, (double)P->flow_grid / (double)(per_cu * cus));
      if (!P->flow) { /* launch-per-depth path */ }
      else if (P->flow_first > 0) {   // children that the launches finish are not waited for
        for (auto& d2 : dep) d2.y = 0;
        for (int i = P->flow_first; i < (int)dep.size(); ++i) if (dep[i].x >= 0) dep[dep[i].x].y++;
      }
      if ((rc = up_to_dev(*P, b.stream, dep, (const int2**)&P->d_dep))) return rc;
      const size_t nints = 3 * dep.size() + 8;
      if ((rc = plan_alloc(&p, nints * sizeof(int)))) return rc;
      P->d_flow = (int*)p;
      SSLAM_HIP_TRY(hipMemsetAsync(p, 0, nints * sizeof(int), b.stream));
      P->flow_epoch = 0; P->spec_epoch = 0; P->lm_epoch = 0;
      // speculative damping trials: one small graph whose ten lanes of pieces are all on the chip at once
      const int K = 10;
      const int spec_mode = chol_spec_mode(b);
      if (spec_mode && b.V.B == 1 && P->flow_launch0 == 0) {   // (the lanes cost three allocations + memsets per rebuild)
        // K lanes of persistent workgroups, all of them on the chip at once.  Lane 0 -- the only one at work until a trial of the iteration
        // has been rejected (adaptive form) -- gets a workgroup per piece like the plain single-launch solve, the others share the rest
        SpecLanes& SL = P->spec;
        int per_cu_s = 0;   // what the device holds at once of the round's own kernel (ten lanes at work must all be resident)
        const void* fn_s = P->ustage ? (const void*)k_chol_spec_round<512, true> : (const void*)k_chol_spec_round<512, false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_s, fn_s, 512, lds_max) != hipSuccess || per_cu_s < 1) per_cu_s = 1;
        const int full = std::max(K, std::min(2 * cap, per_cu_s * cus));
        SL.g0 = std::max(1, std::min((int)dep.size(), std::min(cap, full / 2)));
        SL.g1 = std::max(1, std::min((int)dep.size(), (full - SL.g0) / (K - 1)));
        // adaptive form: only while every lane gets a workgroup per piece -- a lane that walks its pieces in rounds makes the one round of
        // nine trials slower than the trials one after the other (436 keyframes, 121 pieces, 43 workgroups per lane: 7.8 vs 7.5 ms per tick;
        // 110 keyframes, 26 pieces: 5.45 vs 6.00 ms).  A graph that outgrows the lanes goes on with the sequential trials.
        const bool lanes_fit = SL.g1 >= (int)dep.size();
        if (spec_mode == 2 || lanes_fit) {
          SL.after = spec_mode == 2 ? 0 : 1;
          P->spec_grid = SL.g0 + (K - 1) * SL.g1;
          P->spec_need = std::min(1.0, (double)P->spec_grid / (double)(per_cu_s * cus));
          SL.sL = (H.lnz + 64 + 1) & ~1LL; SL.sU = (H.unz + 64 + 1) & ~1LL; SL.sy = (C.dim + 8 + 1) & ~1LL; SL.sx = (C.dim + 8 + 1) & ~1LL;
          SL.spose = (long long)b.V.nPose * 8; SL.slmk = (long long)b.V.nLm * 4; SL.spe = (long long)b.V.B * b.V.maxEdgeChunks; SL.spa = (long long)b.V.B * b.V.maxRowChunks;
          SL.sflow = (long long)nints;
          const size_t nd = (size_t)K * (SL.sL + SL.sU + SL.sy + SL.sx + SL.spose + SL.slmk + SL.spe + SL.spa);
          if ((rc = plan_alloc(&p, nd * sizeof(double)))) return rc;
          SSLAM_HIP_TRY(hipMemsetAsync(p, 0, nd * sizeof(double), b.stream));
          double* dp = (double*)p;
          SL.Lval = dp; dp += K * SL.sL; SL.Uval = dp; dp += K * SL.sU; SL.y = dp; dp += K * SL.sy; SL.x = dp; dp += K * SL.sx;
          SL.pose_trial = dp; dp += K * SL.spose; SL.lmk_trial = dp; dp += K * SL.slmk; SL.part_e = dp; dp += K * SL.spe; SL.part_a = dp;
          const size_t ni = (size_t)K * (SL.sflow + b.V.B);
          if ((rc = plan_alloc(&p, ni * sizeof(int)))) return rc;
          SSLAM_HIP_TRY(hipMemsetAsync(p, 0, ni * sizeof(int), b.stream));
          SL.flow = (int*)p; SL.fail = (int*)p + (size_t)K * SL.sflow;
          if ((rc = plan_alloc(&p, (size_t)K * b.V.B * sizeof(LmState)))) return rc;
          SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (size_t)K * b.V.B * sizeof(LmState), b.stream));
          SL.lm = (LmState*)p;
          SL.pose_cur = b.V.pose; SL.lmk_cur = b.V.lmk;
          if ((rc = plan_alloc(&p, (size_t)(2 * K + 16) * sizeof(int)))) return rc;
          SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (size_t)(2 * K + 16) * sizeof(int), b.stream));
          SL.ctl = (int*)p;
          SL.err = P->d_flow + 3 * dep.size();
          SL.K = K;
        }
      }
    }
  }
  if (P->arena && P->arena->flush(b.stream)) return set_error(SSLAM_ERR_HIP, "upload of the plan tables failed");
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  return 0;
}

// Persistent launches (k_chol_flow, k_chol_spec_round) spin on counters that only workgroups of the SAME launch advance: a launch whose
// grid is not wholly on the chip can wait for a workgroup that has not started (round-4 ADVICE).  Round 5 chained all of them per device,
// one at a time -- correct, but eight robots on one GPU queued behind each other while each used a tenth of the chip.  Round 6: a BUDGET.
// Every plan knows the share of the device its grid occupies when resident (`need`: workgroups / what the device holds of them); the
// ledger keeps, per stream, the need of that stream's latest persistent launch (a stream runs its own launches one after the other).  A new
// launch may overlap the launches of the OTHER streams as long as the shares add up to <= 1; otherwise it waits (an event behind everything
// such a stream holds) for as many of them as it takes, largest first.  Launches enqueued later make the same check against this one, so at
// any time the launches that are not ordered behind one another fit the device.  One stream (the orchestrator) never waits and records no
// event.  A stream whose event has completed is idle and leaves the ledger.
struct PersistEntry { hipStream_t stream; double need; hipEvent_t ev; bool fresh; };   // fresh: ev was recorded behind the stream's latest launch
struct PersistGate {
  std::mutex mu;
  std::vector<PersistEntry> entries;
};
static PersistGate& persist_gate(int device) {
  static std::mutex mu;
  static std::vector<PersistGate*> gates;
  std::lock_guard<std::mutex> lk(mu);
  if ((int)gates.size() <= device) gates.resize(device + 1, nullptr);
  if (!gates[device]) gates[device] = new PersistGate();
  return *gates[device];
}
// Construct before the launch, destroy after it (the lock orders the bookkeeping of concurrent host threads, not the launches).
struct PersistScope {
  PersistGate& g; hipStream_t s; double need; std::unique_lock<std::mutex> lk;
  PersistScope(int device, hipStream_t stream, double need_) : g(persist_gate(device)), s(stream), need(std::min(1.0, std::max(0.0, need_))), lk(g.mu) {
    // idle streams leave the ledger
    for (size_t i = 0; i < g.entries.size();) {
      PersistEntry& e = g.entries[i];
      if (e.stream != s && e.fresh && e.ev && hipEventQuery(e.ev) == hipSuccess) { (void)hipEventDestroy(e.ev); g.entries.erase(g.entries.begin() + i); }
      else ++i;
    }
    double others = 0;
    for (const PersistEntry& e : g.entries) if (e.stream != s) others += e.need;
    std::vector<char> waited(g.entries.size(), 0);
    while (others + need > 1.0 + 1e-9) {
      int pick = -1;
      for (size_t i = 0; i < g.entries.size(); ++i)
        if (g.entries[i].stream != s && !waited[i] && (pick < 0 || g.entries[i].need > g.entries[pick].need)) pick = (int)i;
      if (pick < 0) break;
      PersistEntry& e = g.entries[pick];
      bool chained = false;
      if (!e.ev && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) e.ev = nullptr;
      if (e.ev) {
        if (!e.fresh && hipEventRecord(e.ev, e.stream) == hipSuccess) e.fresh = true;
        if (e.fresh && hipStreamWaitEvent(s, e.ev, 0) == hipSuccess) chained = true;
      }
      if (!chained) (void)hipStreamSynchronize(e.stream);   // (round-5 ADVICE: never let the launch go out unchained)
      waited[pick] = 1;
      others -= e.need;
    }
  }
  ~PersistScope() {
    for (PersistEntry& e : g.entries)
      if (e.stream == s) { e.need = need; e.fresh = false; return; }
    g.entries.push_back(PersistEntry{s, need, nullptr, false});
  }
};
// a stream is about to be destroyed (its owner synchronises it first): no later launch may record an event on it
void persist_forget_stream(int device, hipStream_t stream) {
  PersistGate& g = persist_gate(device);
  std::lock_guard<std::mutex> lk(g.mu);
  for (size_t i = 0; i < g.entries.size(); ++i)
    if (g.entries[i].stream == stream) { if (g.entries[i].ev) (void)hipEventDestroy(g.entries[i].ev); g.entries.erase(g.entries.begin() + i); return; }
}

// LDS bytes the dependency-driven launches reserve: the largest piece of the plan, whichever kernels run it
static size_t plan_flow_lds(const CholPlan& P) {
  size_t lds = (size_t)std::max(std::max(P.tail_lds_f, P.tail_lds_b), P.tail_lds_ff);
  for (size_t l = 0; l < P.plv_lds_f.size(); ++l) lds = std::max(lds, (size_t)std::max(P.plv_lds_f[l], P.plv_lds_b[l]));
  for (int v : P.plv_lds_ff) lds = std::max(lds, (size_t)v);
  return lds * sizeof(double);
}

// (H + lambda I) dx = b in ONE launch (k_chol_flow) for plans that allow it; false: the caller takes the launch-per-depth path
bool chol_plan_flow(const Batch& b) { return b.chol && b.chol->flow && !b.chol->compact; }
// the launches of one solve: per-depth launches over the wide bottom of the tree, the dependency-driven launch over the rest (factor and
// both substitutions), per-depth launches of the backward substitution over the bottom again
static void flow_launches(Batch& b, bool spec = false, bool backward = true, bool lmstep = false, int max_iters = 0) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  const size_t lds = plan_flow_lds(P);
  const int np = (int)P.lp_graph.size();
  for (int l = 0; l < P.flow_launch0; ++l) {
    const int n = P.plv_ptr[l + 1] - P.plv_ptr[l];
    if (P.ustage) hipLaunchKernelGGL((k_chol_pieces<512, true>), dim3(n), dim3(512), (size_t)P.plv_lds_f[l] * sizeof(double), b.stream, b.V, C, P.plv_ptr[l], (const int*)nullptr);
    else hipLaunchKernelGGL((k_chol_pieces<512, false>), dim3(n), dim3(512), (size_t)P.plv_lds_f[l] * sizeof(double), b.stream, b.V, C, P.plv_ptr[l], (const int*)nullptr);
  }
  const int epoch = spec ? ++P.spec_epoch : ++P.flow_epoch;   // the lanes count on their own counters
  const SpecLanes SL = spec ? P.spec : SpecLanes{};
  const int lm_epoch = lmstep ? ++P.lm_epoch : 0;
  const dim3 grid(spec ? P.spec_grid : P.flow_grid);
  const int defer = 2;   // tables + H before the wait for the children (round 4: -0.43 ms per tick)
  {
    PersistScope gate(b.device, b.stream, spec ? P.spec_need : P.flow_need);
    if (P.ustage) hipLaunchKernelGGL((k_chol_flow<512, true>), grid, dim3(512), lds, b.stream, b.V, C, P.flow_first, np, epoch, (const int2*)P.d_dep, P.d_flow, SL, (backward ? 1 : 0) | defer | (lmstep ? 4 : 0), b.d_part_e, max_iters, lm_epoch);
    else hipLaunchKernelGGL((k_chol_flow<512, false>), grid, dim3(512), lds, b.stream, b.V, C, P.flow_first, np, epoch, (const int2*)P.d_dep, P.d_flow, SL, (backward ? 1 : 0) | defer | (lmstep ? 4 : 0), b.d_part_e, max_iters, lm_epoch);
  }
  for (int l = P.flow_launch0 - 1; l >= 0 && backward; --l) {
    const int n = P.plv_ptr[l + 1] - P.plv_ptr[l];
    hipLaunchKernelGGL(k_chol_back_pieces<512>, dim3(n), dim3(512), (size_t)P.plv_lds_b[l] * sizeof(double), b.stream, C, P.plv_ptr[l], (const double*)C.y, b.V.x, (const LmState*)b.V.lm, (const int*)nullptr);
  }
}
int chol_solve_flow(Batch& b) {
  CholPlan& P = *b.chol;
  P.C.flat_L = 0;
  const CholView& C = P.C;
  ScopedTimer t(b, "factor");
  const size_t lds = plan_flow_lds(P);
  const int np = (int)P.lp_graph.size();
  hipLaunchKernelGGL(k_chol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  flow_launches(b);
  hipLaunchKernelGGL(k_chol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky flow launch: %s", hipGetErrorString(e));
  return 0;
}
// the flat factor (multi right-hand-side / path-marginal kernels) through the single launch: L only, no backward substitution
int chol_factor_flat_flow(Batch& b) {
  CholPlan& P = *b.chol;
  P.C.flat_L = 1;
  hipLaunchKernelGGL(k_chol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, P.C);
  flow_launches(b, false, false);
  hipLaunchKernelGGL(k_chol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, P.C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky flow launch: %s", hipGetErrorString(e));
  return 0;
}
// one damping trial of every active graph of a small batch after the Jacobian kernels: begin step, single-launch solve, the rest of the step
int chol_lm_step_flow(Batch& b, int max_iters) {
  CholPlan& P = *b.chol;
  P.C.flat_L = 0;
  const CholView& C = P.C;
  const size_t lds = plan_flow_lds(P);
  const int np = (int)P.lp_graph.size();
  // the begin / end halves of the trial inside the launch when it covers the whole tree and has a workgroup per graph
  if (P.flow_launch0 == 0 && P.flow_grid >= b.V.B) { flow_launches(b, false, true, true, max_iters); }
  else {
    hipLaunchKernelGGL(k_lm_begin_small<512>, dim3(b.V.B), dim3(512), 0, b.stream, b.V, C);
    flow_launches(b);
    hipLaunchKernelGGL(k_lm_end_small<512>, dim3(b.V.B), dim3(512), 0, b.stream, b.V, C, b.d_part_e, max_iters);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "LM step launch: %s", hipGetErrorString(e));
  return 0;
}
// one LM ITERATION of a single small graph: all of its (up to ten) damping trials side by side, then the accept / reject replay
bool chol_plan_spec(const Batch& b) { return chol_plan_flow(b) && b.chol->spec.K > 0; }
int chol_lm_step_spec(Batch& b, int max_iters) {
  CholPlan& P = *b.chol;
  P.C.flat_L = 0;
  const size_t lds = plan_flow_lds(P);
  const int np = (int)P.lp_graph.size();
  const int round = ++P.spec_epoch;
  {
    PersistScope gate(b.device, b.stream, P.spec_need);
    if (P.ustage) hipLaunchKernelGGL((k_chol_spec_round<512, true>), dim3(P.spec_grid), dim3(512), lds, b.stream, b.V, P.C, np, (const int2*)P.d_dep, P.spec, round, max_iters);
    else hipLaunchKernelGGL((k_chol_spec_round<512, false>), dim3(P.spec_grid), dim3(512), lds, b.stream, b.V, P.C, np, (const int2*)P.d_dep, P.spec, round, max_iters);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "speculative LM step launch: %s", hipGetErrorString(e));
  return 0;
}
// the error flag of k_chol_flow (a dependency wait that gave up); call after a stream synchronisation point
int chol_flow_check(Batch& b) {
  if (!b.chol || !b.chol->flow || !b.chol->d_flow) return 0;
  int* d_err = b.chol->d_flow + 3 * b.chol->lp_graph.size();   // the speculative lanes report into the same word (SpecLanes::err)
  int local = 0;
  int* stage = reinterpret_cast<int*>(b.pin->get(sizeof(int)));
  if (!stage) stage = &local;
  SSLAM_HIP_TRY(hipMemcpyAsync(stage, d_err, sizeof(int), hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  if (*stage) {   // reported once: cleared, so that the next solve of the handle is judged on its own
    SSLAM_HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), b.stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
    return set_error(SSLAM_ERR_HIP, "sparse Cholesky: a dependency wait of the single-launch factorisation timed out (the graphs concerned carry a failed trial)");
  }
  return 0;
}

int64_t chol_plan_lnz(const Batch& b) { return b.chol ? b.chol->lnz : 0; }
int chol_plan_levels(const Batch& b) { return b.chol ? b.chol->C.nlevels : 0; }
int chol_plan_launches(const Batch& b) { return b.chol ? (int)b.chol->plv_lds_f.size() + (b.chol->tail_total > 0 ? 1 : 0) : 0; }

// LM endgame: from now on only the graphs flagged in `active` can be in a trial (a terminated graph never comes back within an
// optimize call).  With few of them left, the factor / solve launches are sized for their pieces alone: index lists instead of the
// full piece ranges.  active == nullptr (or most graphs active): back to the full ranges.
int chol_set_active(Batch& b, const std::vector<char>* active) {
  if (!b.chol) return 0;
  CholPlan& P = *b.chol;
  P.compact = false;
  if (!active) return 0;
  const int B = b.V.B;
  int na = 0;
  for (int g = 0; g < B; ++g) na += (*active)[g] ? 1 : 0;
  if (B < 8 || na == 0 || 2 * na > B) return 0;
  const int nplv = (int)P.plv_lds_f.size();
  std::vector<int> idx;
  idx.reserve(P.lp_graph.size() * (size_t)na / B + 64);
  P.c_ptr.assign(nplv + 2, 0);
  for (int l = 0; l < nplv; ++l) {
    P.c_ptr[l] = (int)idx.size();
    for (int q = P.plv_ptr[l]; q < P.plv_ptr[l + 1]; ++q) if ((*active)[P.lp_graph[q]]) idx.push_back(q);
  }
  P.c_ptr[nplv] = (int)idx.size();
  if (P.tail_total > 0) for (int g = 0; g < B; ++g) if ((*active)[g]) idx.push_back(g);   // one tail workgroup per graph (a graph without tail pieces returns at once)
  P.c_ptr[nplv + 1] = (int)idx.size();
  if (idx.size() > P.idx_cap) {
    if (P.d_idx) (void)hipFree(P.d_idx);
    P.d_idx = nullptr; P.idx_cap = 0;
    SSLAM_HIP_TRY(hipMalloc((void**)&P.d_idx, (idx.size() + 1024) * sizeof(int)));
    P.idx_cap = idx.size() + 1024;
  }
  if (!idx.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(P.d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));   // idx is a local
  P.compact = true;
  return 0;
}

int chol_factor_and_forward(Batch& b, bool flat) {
  CholPlan& P = *b.chol;
  P.C.flat_L = flat ? 1 : 0;   // form of the factor in HBM: flat for the multi right-hand-side kernels, class-interleaved for chol_backward
  const CholView& C = P.C;
  ScopedTimer t(b, "factor");
  hipLaunchKernelGGL(k_chol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  const int nplv = (int)P.plv_lds_f.size();
  for (int l = 0; l < nplv; ++l) {
    const int n = P.compact ? P.c_ptr[l + 1] - P.c_ptr[l] : P.plv_ptr[l + 1] - P.plv_ptr[l];
    if (n <= 0) continue;
    const int* idx = P.compact ? P.d_idx + P.c_ptr[l] : nullptr;
    if (P.front) {   // front tables: one kernel for every class, the workgroup size the launch was cut for
      const size_t ldf = (size_t)P.plv_lds_ff[l] * sizeof(double);
#define SSLAM_LAUNCH_FRONT(NTV)                                                                                                     \
  if (P.plv_cls[l] == 1) hipLaunchKernelGGL((k_front_pieces<NTV, true>), dim3(n), dim3(NTV), ldf, b.stream, b.V, C, P.plv_ptr[l], idx);   \
  else hipLaunchKernelGGL((k_front_pieces<NTV, false>), dim3(n), dim3(NTV), ldf, b.stream, b.V, C, P.plv_ptr[l], idx);
      switch (P.plv_nt[l]) {
        case 128: SSLAM_LAUNCH_FRONT(128) break;
        case 256: SSLAM_LAUNCH_FRONT(256) break;
        case 512: SSLAM_LAUNCH_FRONT(512) break;
        case 1024: SSLAM_LAUNCH_FRONT(1024) break;
        default: SSLAM_LAUNCH_FRONT(64) break;
      }
#undef SSLAM_LAUNCH_FRONT
      continue;
    }
    const size_t lds = (size_t)P.plv_lds_f[l] * sizeof(double);
#define SSLAM_LAUNCH_PIECES(NTV)                                                                                                   \
  if (P.ustage) hipLaunchKernelGGL((k_chol_pieces<NTV, true>), dim3(n), dim3(NTV), lds, b.stream, b.V, C, P.plv_ptr[l], idx);       \
  else hipLaunchKernelGGL((k_chol_pieces<NTV, false>), dim3(n), dim3(NTV), lds, b.stream, b.V, C, P.plv_ptr[l], idx);
#define SSLAM_LAUNCH_MID(NTV)                                                                                                      \
  if (C.rupd) hipLaunchKernelGGL((k_chol_pieces<NTV, false, true>), dim3(n), dim3(NTV), lds, b.stream, b.V, C, P.plv_ptr[l], idx);  \
  else hipLaunchKernelGGL((k_chol_pieces<NTV, false, false>), dim3(n), dim3(NTV), lds, b.stream, b.V, C, P.plv_ptr[l], idx);
    if (P.plv_cls[l] == 1) {   // mid pieces: wider workgroups, right-looking internal updates, update-matrix records from HBM
      switch (P.plv_nt[l]) {
        case 128: SSLAM_LAUNCH_MID(128) break;
        case 512: SSLAM_LAUNCH_MID(512) break;
        default: SSLAM_LAUNCH_MID(256) break;
      }
    } else switch (P.nt_leaf) {
      case 128: SSLAM_LAUNCH_PIECES(128) break;
      case 512: SSLAM_LAUNCH_PIECES(512) break;
      case 1024: SSLAM_LAUNCH_PIECES(1024) break;
      case 256: SSLAM_LAUNCH_PIECES(256) break;
      default: SSLAM_LAUNCH_PIECES(64) break;
    }
#undef SSLAM_LAUNCH_PIECES
#undef SSLAM_LAUNCH_MID
  }
  if (P.tail_total > 0) {
    const int n = P.compact ? P.c_ptr[nplv + 1] - P.c_ptr[nplv] : b.V.B;
    const int* idx = P.compact ? P.d_idx + P.c_ptr[nplv] : nullptr;
    if (n > 0 && P.front) {
      const int ntf = P.nt_tail == 1024 ? 1024 : P.nt_ftail;
      if (ntf == 1024) hipLaunchKernelGGL(k_front_tail<1024>, dim3(n), dim3(1024), (size_t)P.tail_lds_ff * sizeof(double), b.stream, b.V, C, idx);
      else if (ntf == 128) hipLaunchKernelGGL(k_front_tail<128>, dim3(n), dim3(128), (size_t)P.tail_lds_ff * sizeof(double), b.stream, b.V, C, idx);
      else if (ntf == 256) hipLaunchKernelGGL(k_front_tail<256>, dim3(n), dim3(256), (size_t)P.tail_lds_ff * sizeof(double), b.stream, b.V, C, idx);
      else hipLaunchKernelGGL(k_front_tail<512>, dim3(n), dim3(512), (size_t)P.tail_lds_ff * sizeof(double), b.stream, b.V, C, idx);
    } else if (n > 0) {
      if (P.nt_tail == 1024) hipLaunchKernelGGL(k_chol_tail<1024>, dim3(n), dim3(1024), (size_t)P.tail_lds_f * sizeof(double), b.stream, b.V, C, idx);
      else hipLaunchKernelGGL(k_chol_tail<512>, dim3(n), dim3(512), (size_t)P.tail_lds_f * sizeof(double), b.stream, b.V, C, idx);
    }
  }
  hipLaunchKernelGGL(k_chol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky factor launch: %s", hipGetErrorString(e));
  return 0;
}

int chol_backward(Batch& b) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  if (C.flat_L) return set_error(SSLAM_ERR_INVALID, "chol_backward needs the class-interleaved factor (the last factorisation was a flat one)");
  ScopedTimer t(b, "solve");
  const int nplv = (int)P.plv_lds_b.size();
  if (P.tail_total > 0) {
    const int n = P.compact ? P.c_ptr[nplv + 1] - P.c_ptr[nplv] : b.V.B;
    const int* idx = P.compact ? P.d_idx + P.c_ptr[nplv] : nullptr;
    const int ntb = P.nt_btail == 128 || P.nt_btail == 256 ? P.nt_btail : 512;
    if (n > 0 && ntb == 128) hipLaunchKernelGGL(k_chol_back_tail<128>, dim3(n), dim3(128), (size_t)P.tail_lds_b * sizeof(double), b.stream, C, (const double*)C.y, b.V.x, (const LmState*)b.V.lm, idx);
    else if (n > 0 && ntb == 256) hipLaunchKernelGGL(k_chol_back_tail<256>, dim3(n), dim3(256), (size_t)P.tail_lds_b * sizeof(double), b.stream, C, (const double*)C.y, b.V.x, (const LmState*)b.V.lm, idx);
    else if (n > 0) hipLaunchKernelGGL(k_chol_back_tail<512>, dim3(n), dim3(512), (size_t)P.tail_lds_b * sizeof(double), b.stream, C, (const double*)C.y, b.V.x, (const LmState*)b.V.lm, idx);
  }
  for (int l = nplv - 1; l >= 0; --l) {
    const int n = P.compact ? P.c_ptr[l + 1] - P.c_ptr[l] : P.plv_ptr[l + 1] - P.plv_ptr[l];
    if (n <= 0) continue;
    const int* idx = P.compact ? P.d_idx + P.c_ptr[l] : nullptr;
    const size_t lds = (size_t)P.plv_lds_b[l] * sizeof(double);
#define SSLAM_LAUNCH_BACK(NTV) hipLaunchKernelGGL(k_chol_back_pieces<NTV>, dim3(n), dim3(NTV), lds, b.stream, C, P.plv_ptr[l], (const double*)C.y, b.V.x, (const LmState*)b.V.lm, idx);
    const int ovr = P.plv_cls[l] == 1 ? P.nt_bmid : P.nt_bleaf;
    switch (ovr > 0 ? ovr : P.plv_nt[l]) {
      case 128: SSLAM_LAUNCH_BACK(128) break;
      case 512: SSLAM_LAUNCH_BACK(512) break;
      case 1024: SSLAM_LAUNCH_BACK(1024) break;
      case 256: SSLAM_LAUNCH_BACK(256) break;
      default: SSLAM_LAUNCH_BACK(64) break;
    }
#undef SSLAM_LAUNCH_BACK
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky solve launch: %s", hipGetErrorString(e));
  return 0;
}

// X = (L L^T)^-1 RHS for nrhs right-hand sides (host arrays, internal ordering, [nrhs][dim])
int chol_solve_multi(Batch& b, const double* rhs_host, int nrhs, double* x_host) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  if (nrhs <= 0) return 0;
  if (!C.flat_L) return set_error(SSLAM_ERR_INVALID, "chol_solve_multi needs the flat factor (chol_factor_and_forward(b, true))");
  const int chunk_max = std::max(1, std::min(nrhs, (int)std::min<int64_t>(4096, ((int64_t)1 << 30) / std::max(1, C.dim) / 8)));
  if (P.multi_cap < chunk_max) {
    if (P.d_multi_y) (void)hipFree(P.d_multi_y);
    if (P.d_multi_x) (void)hipFree(P.d_multi_x);
    SSLAM_HIP_TRY(hipMalloc((void**)&P.d_multi_y, (size_t)chunk_max * C.dim * sizeof(double)));
    SSLAM_HIP_TRY(hipMalloc((void**)&P.d_multi_x, (size_t)chunk_max * C.dim * sizeof(double)));
    P.multi_cap = chunk_max;
  }
  for (int r0 = 0; r0 < nrhs; r0 += chunk_max) {
    const int nr = std::min(chunk_max, nrhs - r0);
    const size_t bytes = (size_t)nr * C.dim * sizeof(double);
    SSLAM_HIP_TRY(hipMemcpyAsync(P.d_multi_x, rhs_host + (size_t)r0 * C.dim, bytes, hipMemcpyHostToDevice, b.stream));
    for (int l = 0; l < C.nlevels; ++l) {
      const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
      if (n > 0) hipLaunchKernelGGL(k_chol_forward_level, dim3(n, nr), dim3(64), 0, b.stream, C, P.lvl_ptr[l], (const double*)P.d_multi_x, P.d_multi_y);
    }
    for (int l = C.nlevels - 1; l >= 0; --l) {
      const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
      if (n > 0) hipLaunchKernelGGL(k_chol_backward_level<1>, dim3((n + 7) / 8, nr), dim3(64), 0, b.stream, C, P.lvl_ptr[l], n, (const double*)P.d_multi_y, P.d_multi_x);
    }
    SSLAM_HIP_TRY(hipMemcpyAsync(x_host + (size_t)r0 * C.dim, P.d_multi_x, bytes, hipMemcpyDeviceToHost, b.stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  }
  return 0;
}

// Z(v, v) of (L L^T)^-1 for the vertices whose unknowns start at xoff[k] (internal row order, dims[k] = 3 | 6), from the last FLAT
// factorisation; out: [n][36], the leading dims[k]^2 entries row-major.  Returns SSLAM_ERR_UNSUPPORTED when a path does not fit the LDS of
// one wave (the caller falls back to the multi right-hand-side solves).
int chol_marginal_diag(Batch& b, const std::vector<int>& xoff, const std::vector<int>& dims, double* out) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  const int n = (int)xoff.size();
  if (n == 0) return 0;
  if (!C.flat_L) return set_error(SSLAM_ERR_INVALID, "chol_marginal_diag needs the flat factor (chol_factor_and_forward(b, true))");
  std::vector<int> hdr(2 * (size_t)n + 1, 0), cols;
  int maxlen = 1;
  for (int k = 0; k < n; ++k) {
    int j = (xoff[k] >= 0 && xoff[k] < (int)P.h_xoff_col.size()) ? P.h_xoff_col[xoff[k]] : -1;
    if (j < 0) return set_error(SSLAM_ERR_INVALID, "marginal of an unknown that is not the start of a block row");
    hdr[k] = (int)cols.size();
    for (; j >= 0; j = P.h_cparent[j]) cols.push_back(j);
    maxlen = std::max(maxlen, (int)cols.size() - hdr[k]);
    hdr[n + 1 + k] = dims[k];
  }
  hdr[n] = (int)cols.size();
  const size_t lds = (size_t)maxlen * 36 * sizeof(double) + 36 * sizeof(double) + (size_t)maxlen * (sizeof(int4) + sizeof(int)) + 16;
  if (lds > 60 * 1024) return SSLAM_ERR_UNSUPPORTED;
  const size_t ints = hdr.size() + cols.size();
  // scratch: out of the graph handle's arena when there is one (a plan is rebuilt every tick of the orchestrator: no hipMalloc / hipFree
  // pair per tick), else the plan's own allocations
  if (P.mpath_cap < ints) {
    if (P.arena) { P.d_mpath = (int*)P.arena->take((ints + 1024) * sizeof(int), true); if (!P.d_mpath) return set_error(SSLAM_ERR_HIP, "device allocation failed"); }
    else { if (P.d_mpath) (void)hipFree(P.d_mpath); P.d_mpath = nullptr; P.mpath_cap = 0; SSLAM_HIP_TRY(hipMalloc((void**)&P.d_mpath, (ints + 1024) * sizeof(int))); }
    P.mpath_cap = ints + 1024;
  }
  if (P.mout_cap < (size_t)n * 36) {
    if (P.arena) { P.d_mout = (double*)P.arena->take(((size_t)n * 36 + 1024) * sizeof(double), true); if (!P.d_mout) return set_error(SSLAM_ERR_HIP, "device allocation failed"); }
    else { if (P.d_mout) (void)hipFree(P.d_mout); P.d_mout = nullptr; P.mout_cap = 0; SSLAM_HIP_TRY(hipMalloc((void**)&P.d_mout, ((size_t)n * 36 + 1024) * sizeof(double))); }
    P.mout_cap = (size_t)n * 36 + 1024;
  }
  hdr.insert(hdr.end(), cols.begin(), cols.end());
  // both copies through the page-locked staging buffer: [paths | results]
  const size_t in_bytes = (hdr.size() * sizeof(int) + 7) & ~(size_t)7, out_bytes = (size_t)n * 36 * sizeof(double);
  char* stage = b.pin->get(in_bytes + out_bytes);
  const void* src = hdr.data();
  void* dst = out;
  if (stage) { memcpy(stage, hdr.data(), hdr.size() * sizeof(int)); src = stage; dst = stage + in_bytes; }
  SSLAM_HIP_TRY(hipMemcpyAsync(P.d_mpath, src, hdr.size() * sizeof(int), hipMemcpyHostToDevice, b.stream));
  hipLaunchKernelGGL(k_chol_marginal_paths, dim3(n), dim3(64), lds, b.stream, C, (const int*)P.d_mpath, (const int*)(P.d_mpath + n + 1),
                     (const int*)(P.d_mpath + 2 * (size_t)n + 1), P.d_mout, maxlen);
  SSLAM_HIP_TRY(hipMemcpyAsync(dst, P.d_mout, out_bytes, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));   // hdr is a local
  if (stage) memcpy(out, stage + in_bytes, out_bytes);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "path marginals launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace sslam
