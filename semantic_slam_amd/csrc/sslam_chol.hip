// Sparse block Cholesky for the LM normal equations on gfx950 (FP64).
//
// Replaces g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with
// "lm_var" (reference src/ps_graph_slam/graph_slam.cpp:27,67-73): fill-reducing
// ordering on the BLOCK pattern, symbolic factorisation once per structure, numeric
// factorisation every LM trial, triangular solves.  MI355X design:
//   * symbolic phase on the host (minimum degree with explicit fill on the 6x6/3x3 block graph,
//     elimination tree levels, per-target-block update lists),
//   * numeric phase on the device, left-looking *gather* form (every L block is written by exactly one
//     workgroup -> deterministic, no atomics):
//       - wide levels of the elimination tree: one launch per level, one workgroup (1 or 4 waves) per
//         block column (k_chol_level<64/256>, flat update ranges staged through LDS),
//       - narrow levels: 16 waves per column, update lists cut into <= 256 items of 4-lane register
//         tiles (k_chol_level<1024>),
//       - the top of the tree (levels a few columns wide): ONE launch, one workgroup per graph walking
//         supernodes (chains with nested structure) through an LDS panel (k_chol_tail),
//     forward substitution fused into the factorisation (b is carried as an extra block row),
//     backward substitution as a top-down sweep with the same level / head split,
//   * all graphs of a batch share the launches (levels are concatenated across graphs).
// Tuning knobs read from the environment when a plan is built (defaults in parentheses):
//   SSLAM_CHOL_TAIL_WIDTH (6 for batches of >= 32 graphs, else 2; 0 = no tail kernel),
//   SSLAM_CHOL_SUPERNODE (6 = kMaxSn columns per supernode; 1 = singletons), SSLAM_CHOL_DUMP (level / supernode statistics).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <vector>

#include "../../include/sslam.h"
#include "graph_engine.hpp"

namespace sslam {

// Packed per-column / per-block / per-update records: one 16/32-byte load each instead of a chain of
// dependent 4-byte loads (a narrow level is a single workgroup whose run time is that chain).
struct ColMeta { int xoff, dim, graph, b0, nb, base, csize, ubase, ucount, ibase, icount, chunk; };  // base = Lval offset of the diagonal block;
                                                     // [ubase, ubase + ucount) = the column's updates,
                                                     // [ibase, ibase + icount) = its work items of <= chunk updates each
struct BlkMeta { int off, di, src, fmt, up0, up1, rowcol, xoff_row, it0, nit; };   // src: H offset or -1; rowcol: column id of the
                                                                                   // row; [it0, it0 + nit) = items of this block
struct UpdMeta { int ua, ub, ux, pk; };                                  // offsets of L_ik, L_jk, y_k; pk = target descriptor:
constexpr int kMaxSn = 6;               // columns per supernode
struct SnMeta { int q0, s, ibase, icount, chunk, lds, poff[kMaxSn + 2]; };   // supernode = tail columns [q0, q0 + s) (a chain with
                                       // nested structure), its external-update items [ibase, ibase + icount), LDS doubles needed,
                                       // LDS panel offset of every column (poff[s] = panel size)
struct ItemMeta { UpdMeta first; int u0, n, pad0, pad1; };               // <= chunk consecutive updates of ONE target block; the
                                                                         // first update rides along (one dependent load less)
constexpr int kUpdToffMask = 0xFFFFF;   // pk bits 0..19: offset of the target block inside its column
constexpr int kUpdDi6 = 1 << 20;        // target block has 6 rows (else 3)
constexpr int kUpdDk6 = 1 << 21;        // source column k is 6 wide (else 3)
constexpr int kUpdDiag = 1 << 22;       // target is the diagonal block (carries the forward-substitution rhs too)
constexpr int kUpdDj6 = 1 << 23;        // target column is 6 wide (else 3)

struct CholView {
  int ncol, nlevels, dim;
  const ColMeta* col;    // [ncol]
  const BlkMeta* blk;    // [nblk] (blocks of a column are consecutive, diagonal first)
  const UpdMeta* upd;    // update lists, concatenated in block order
  const ItemMeta* item;  // work items, concatenated in block order
  const int* lvl_cols;   // columns grouped by level (within a level: level-scheduled columns first, tail columns last)
  const ColMeta* lcol;   // col[lvl_cols[.]]: the level kernels start from one load instead of two dependent ones
  const int* tail_ptr;   // [B + 1] per graph: its columns factored by k_chol_tail, in elimination order
  const int* tail_cols;
  const int* sn_ptr;     // [B + 1] supernodes of each graph's tail, in elimination order
  const SnMeta* sn;
  double* Lval;
  double* y;             // forward-substituted rhs [dim]
  int* fail;             // [B]
};

struct CholPlan {
  CholView C{};
  std::vector<int> lvl_ptr;
  std::vector<int> lvl_nfactor;  // columns of the level that the level launches factor (the rest belong to a tail)
  int tail_maxEt = 0, tail_total = 0;
  std::vector<int> lvl_maxlist;  // longest update list among the level's blocks
  std::vector<int> lvl_maxEt;    // largest column (entries + rhs) of the level
  std::vector<int> lvl_maxItemLds;  // most LDS doubles a column of the level needs under the item scheme; tail_maxEt likewise
  std::vector<void*> allocs;
  int max_col_entries = 0;
  int64_t lnz = 0;
  double* d_multi_y = nullptr;  // scratch for multi-rhs solves
  double* d_multi_x = nullptr;
  int multi_cap = 0;
};

void chol_plan_free(CholPlan* p) {
  if (!p) return;
  for (void* a : p->allocs) (void)hipFree(a);
  if (p->d_multi_y) (void)hipFree(p->d_multi_y);
  if (p->d_multi_x) (void)hipFree(p->d_multi_x);
  delete p;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// tail of a column: wave 0 factors the D x D diagonal block in registers (every lane the same copy) and
// forward-substitutes the rhs; the other waves pick L_jj and the reciprocal pivots up from LDS and one thread
// per off-diagonal row solves x L_jj^T = v.  Pivots are inverted once (rsqrt) and multiplied from then on:
// FP64 sqrt / divide sequences are what the leaf levels of a batch are bound by.
// KEEP: the factored column (L_jj, the solved rows, y_j) is also written back into `sm` for the in-LDS updates of a supernode
template <int D, int NT, bool KEEP = false>
__device__ __forceinline__ void chol_tail(double* sm, int csize, double* fac, double* Lw, double* yout, int* fail, int tid) {
  double a[D * D], inv[D];
  if (tid < 64) {
    double t[D];
#pragma unroll
    for (int q = 0; q < D * D; ++q) a[q] = sm[q];
#pragma unroll
    for (int q = 0; q < D; ++q) t[q] = sm[csize + q];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double d = a[c * D + c];
#pragma unroll
      for (int s = 0; s < c; ++s) d -= a[c * D + s] * a[c * D + s];
      if (!(d > 0)) { ok = false; d = 1.0; }
      const double id = rsqrt(d);
      inv[c] = id;
      a[c * D + c] = d * id;
#pragma unroll
      for (int r = c + 1; r < D; ++r) {
        double x = a[r * D + c];
#pragma unroll
        for (int s = 0; s < c; ++s) x -= a[r * D + s] * a[c * D + s];
        a[r * D + c] = x * id;
      }
#pragma unroll
      for (int s = c + 1; s < D; ++s) a[c * D + s] = 0.0;  // strict upper = 0
    }
    if (tid == 0) {
      if (!ok) *fail = 1;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        double v = t[r];
#pragma unroll
        for (int s = 0; s < r; ++s) v -= a[r * D + s] * t[s];
        t[r] = v * inv[r];
      }
#pragma unroll
      for (int q = 0; q < D * D; ++q) Lw[q] = a[q];
#pragma unroll
      for (int q = 0; q < D; ++q) yout[q] = t[q];
      if (KEEP) {
#pragma unroll
        for (int q = 0; q < D * D; ++q) sm[q] = a[q];
#pragma unroll
        for (int q = 0; q < D; ++q) sm[csize + q] = t[q];
      }
      if (NT > 64) {
#pragma unroll
        for (int q = 0; q < D * D; ++q) fac[q] = a[q];
#pragma unroll
        for (int q = 0; q < D; ++q) fac[D * D + q] = inv[q];
      }
    }
  }
  const int nrows_off = (csize - D * D) / D;
  if (NT > 64) {
    if (nrows_off <= 0) return;   // uniform per workgroup
    __syncthreads();
    if (tid >= 64 && tid < nrows_off) {
#pragma unroll
      for (int q = 0; q < D * D; ++q) a[q] = fac[q];
#pragma unroll
      for (int q = 0; q < D; ++q) inv[q] = fac[D * D + q];
    }
  }
  for (int row = tid; row < nrows_off; row += NT) {
    const double* v = sm + D * D + row * D;
    double x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double w = v[c];
#pragma unroll
      for (int s = 0; s < c; ++s) w -= x[s] * a[c * D + s];
      x[c] = w * inv[c];
    }
    double* o = Lw + D * D + row * D;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = x[c];
    if (KEEP) {
      double* w = sm + D * D + row * D;
#pragma unroll
      for (int c = 0; c < D; ++c) w[c] = x[c];
    }
  }
}

// One workgroup per block column j of the current level (NT = 64 leaf levels, 256 wide levels, 1024 narrow
// top levels, which are a single column whose run time is a chain of dependent memory latencies):
//   S = A(:,j) + lambda I - sum_k L(:,k) L(j,k)^T,   L(j,j) = chol(S(j,j)),   L(i,j) = S(i,j) L(j,j)^-T,
//   y(j) = L(j,j)^-1 (b(j) - sum_k L(j,k) y(k))                      (forward substitution fused)
// The updates of a column are cut (on the host) into work items of <= chunk consecutive updates of one target
// block, so that a 1024-thread workgroup has an item for each of its 256 four-lane slots.  A slot keeps the
// 3 x 3 tiles of the target in registers (lane = tile (tr, tc)): per update each lane reads its 3 rows of L_ik
// and of L_jk straight from global memory (contiguous 3 dk doubles each) and does 9 dk FMAs -- no LDS staging,
// no cross-lane traffic.  The item's tile goes to its own LDS slot; the slots of a block are summed in item
// order -> deterministic.
constexpr int kItemDoubles = 42;   // LDS doubles per item: 6 x 6 tile entries + 6 rhs components
constexpr int kItemsPerColumn = 256;
// Wide levels (NT < 1024: many columns in flight, short lists) use the other update scheme instead: the column's
// updates form one flat list sorted by target block and wave w owns a contiguous range of it.  The range's
// records are fetched 64 at a time (one per lane) and broadcast with readlane; per update the wave parks L_ik,
// L_jk and y_k in its LDS scratch and the di x dj entry lanes read rows from LDS (broadcast).  When the target
// changes the wave flushes its partial block to its own LDS copy of the column; the copies are reduced in wave
// order.  Fewer registers (occupancy) and no redundant tile loads, but LDS-bandwidth bound on long lists.
constexpr int kPartDoubles = 12288;  // LDS budget for the per-wave partial columns
constexpr int kScr = 104;             // doubles of LDS scratch per wave: L_ik at 0, L_jk at 36 (64 lanes written), y_k at 72
template <int NT>
__device__ __forceinline__ void chol_column(const BatchView& V, const CholView& C, const ColMeta& cm, double* sm) {
  // ITEMS: sm = [Et] column + rhs entries | [icount][kItemDoubles] item partials (the first one doubles as L_jj staging)
  // else:  sm = [Et] column + rhs entries | [nparts][Et] partial columns | [NW][kScr] wave scratch
  constexpr int NW = NT / 64;
  constexpr bool ITEMS = NT >= 1024;
  const int g = cm.graph;
  const int in_trial = V.lm[g].in_trial;   // checked after the update phase: its latency overlaps the source loads
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave-uniform -> scalar metadata loads
  const int dj = cm.dim, b0 = cm.b0, nb = cm.nb, base = cm.base, csize = cm.csize;
  const int Et = csize + dj;  // the last dj "entries" are the forward-substitution rhs
  const double lambda = V.lm[g].lambda;
  const double* __restrict__ H = V.Hpp_diag;
  const double* __restrict__ L = C.Lval;
  const double* __restrict__ Y = C.y;
  double* part = sm + Et;
  const int r = lane / dj, c = lane - r * dj;          // entry lanes: lane < di * dj
  const int ry = lane - 40;                            // rhs lanes: 40 .. 40 + dj - 1 (diagonal block only)
  const int U = cm.ucount;
  const int nparts = max(1, min(min(NW, (U + 7) >> 3), kPartDoubles / Et));
  const int per = (U + nparts - 1) / nparts;
  if (ITEMS) {
    const int tr = (lane >> 1) & 1, tc = lane & 1;
    const int icount = cm.icount, chunk = cm.chunk;
    for (int it0 = wave * 16; it0 < icount; it0 += NT / 4) {     // wave-uniform bound
      const int it = it0 + (lane >> 2);
      const bool have = it < icount;
      const ItemMeta im = C.item[cm.ibase + min(it, icount - 1)];
      const int tpk = im.first.pk;
      const int di = (tpk & kUpdDi6) ? 6 : 3, dj = (tpk & kUpdDj6) ? 6 : 3;
      const bool diag = tpk & kUpdDiag;
      const bool tile = have && 3 * tr < di && 3 * tc < dj;
      const int tre = 3 * tr < di ? tr : 0, tce = 3 * tc < dj ? tc : 0;   // idle lanes shadow tile (0, 0): valid addresses
      double acc[9], accy[3];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) accy[q] = 0;
      UpdMeta um = im.first;
      for (int k = 0; k < chunk; ++k) {
        const bool live = k < im.n;
        const UpdMeta nx = C.upd[im.u0 + min(k + 1, im.n - 1)];    // next update of the item (re-reads the last one at the end)
        const int dk = (um.pk & kUpdDk6) ? 6 : 3;
        const double* A = L + um.ua + 3 * tre * dk;
        const double* B = L + um.ub + 3 * tce * dk;
        const double* yk = Y + um.ux;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // K in halves of 3 (one half when the source column is 3 wide): 21 live doubles, no spills
          const bool on = live && 3 * h < dk;
          const int ko = 3 * h < dk ? 3 * h : 0;
          double a[9], bb[9], yv[3];
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int q = 0; q < 3; ++q) { a[rr * 3 + q] = A[rr * dk + ko + q]; bb[rr * 3 + q] = B[rr * dk + ko + q]; }
#pragma unroll
          for (int q = 0; q < 3; ++q) yv[q] = yk[ko + q];
          if (on) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) acc[rr * 3 + cc] += a[rr * 3] * bb[cc * 3] + a[rr * 3 + 1] * bb[cc * 3 + 1] + a[rr * 3 + 2] * bb[cc * 3 + 2];
              accy[rr] += a[rr * 3] * yv[0] + a[rr * 3 + 1] * yv[1] + a[rr * 3 + 2] * yv[2];
            }
          }
        }
        um = nx;
      }
      if (tile) {
        double* o = part + it * kItemDoubles;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
          if (diag && tc == 0) o[36 + 3 * tr + rr] = accy[rr];
        }
      }
    }
  } else {
    double* scr = sm + Et + nparts * Et + wave * kScr;
    if (wave < nparts && wave * per < U) {
      // Per update the wave loads L_ik, L_jk and y_k with three uniform-base loads (idle lanes re-read the last
      // element: no branches around loads, or the compiler waits for vmcnt(0) at every use and the ring
      // serialises), parks them at fixed LDS offsets 0 / 36 / 72 and every lane forms one dot product
      // row(A) . row(B or y).  Lane roles (rows, flush slot) only change when the target block changes.
      const int lo = wave * per, hi = min(U, lo + per);
      double* mypart = part + wave * Et;
      const int lo36 = min(lane, 35), lo6 = min(lane, 5);
      double acc = 0;
      int cur = -1;            // descriptor of the target being accumulated
      bool active = false;     // lane owns an entry (or rhs component) of the target
      int a3 = 0, a6 = 0, b3 = 0, b6 = 0, foff = 0;
      for (int t0 = lo; t0 < hi; t0 += 64) {
        const int nt = min(64, hi - t0);
        const UpdMeta um = C.upd[cm.ubase + t0 + min(lane, nt - 1)];
        const int mua = um.ua, mub = um.ub, mux = um.ux, mpk = um.pk;
        constexpr int kRing = NT >= 1024 ? 4 : 2;   // short lists on the wide levels: registers buy occupancy there
        double va[kRing], vb[kRing], vy[kRing];
        int pks[kRing];
#define SSLAM_CHOL_ISSUE(q, t)                                                                        \
    {                                                                                                   \
      const int tt_ = min((t), nt - 1);                                                                 \
      const double* A_ = L + __builtin_amdgcn_readlane(mua, tt_);                                       \
      const double* B_ = L + __builtin_amdgcn_readlane(mub, tt_);                                       \
      const double* Y_ = Y + __builtin_amdgcn_readlane(mux, tt_);                                       \
      pks[q] = __builtin_amdgcn_readlane(mpk, tt_);                                                     \
      va[q] = A_[lo36];                                                                                 \
      vb[q] = B_[lo36];                                                                                 \
      vy[q] = Y_[lo6];                                                                                  \
    }
#pragma unroll
        for (int q = 0; q < kRing; ++q) SSLAM_CHOL_ISSUE(q, q)
        for (int t = 0; t < nt; t += kRing) {
#pragma unroll
          for (int q = 0; q < kRing; ++q) {
            const bool live = t + q < nt;
            const int pk = pks[q];
            if (live) {
              if ((pk & kUpdToffMask) != (cur & kUpdToffMask) || cur < 0) {
                if (cur >= 0 && active) mypart[foff] = acc;
                cur = pk; acc = 0;
                const int nE = ((pk & kUpdDi6) ? 6 : 3) * dj;
                const bool isrhs = lane >= nE;
                active = !isrhs || ((pk & kUpdDiag) && ry >= 0 && ry < dj);
                const int row = isrhs ? max(min(ry, 5), 0) : min(r, 5);
                a3 = row * 3; a6 = row * 6;
                b3 = isrhs ? 72 : 36 + c * 3; b6 = isrhs ? 72 : 36 + c * 6;
                foff = isrhs ? csize + ry : (pk & kUpdToffMask) + lane;
              }
              scr[lane] = va[q];
              scr[36 + lane] = vb[q];
              if (lane < 6) scr[72 + lane] = vy[q];
            }
            SSLAM_CHOL_ISSUE(q, t + q + kRing)
            if (live) {
              if (pk & kUpdDk6) {
                const double* pa = scr + a6;
                const double* pb = scr + b6;
                acc += pa[0] * pb[0] + pa[1] * pb[1] + pa[2] * pb[2] + pa[3] * pb[3] + pa[4] * pb[4] + pa[5] * pb[5];
              } else {
                const double* pa = scr + a3;
                const double* pb = scr + b3;
                acc += pa[0] * pb[0] + pa[1] * pb[1] + pa[2] * pb[2];
              }
            }
          }
        }
#undef SSLAM_CHOL_ISSUE
      }
      if (cur >= 0 && active) mypart[foff] = acc;
    }
  }
  // A(:,j) + lambda I and the rhs, one wave per block (lanes as above).  The loads of the first two blocks
  // of every wave are issued before the barrier so that their latency overlaps the update phase of the others.
  auto gather = [&](const BlkMeta& bm, bool diag) -> double {
    double v = 0;
    if (lane < bm.di * dj) {
      if (bm.src >= 0) v = bm.fmt ? H[bm.src + c * bm.di + r] : H[bm.src + r * dj + c];
      if (diag && r == c) v += lambda;
    } else if (diag && ry >= 0 && ry < dj) {
      v = V.bvec[cm.xoff + ry];
    }
    return v;
  };
  double av0 = 0, av1 = 0;
  if (wave < nb) av0 = gather(C.blk[b0 + wave], wave == 0);
  if (wave + NW < nb) av1 = gather(C.blk[b0 + wave + NW], false);
  if (!in_trial) return;   // uniform per workgroup; nothing has been written yet
  __syncthreads();
  for (int bi = wave, it = 0; bi < nb; bi += NW, ++it) {
    const BlkMeta bm = C.blk[b0 + bi];
    const bool diag = (bi == 0);
    double v = it == 0 ? av0 : (it == 1 ? av1 : gather(bm, diag));
    const bool ent = lane < bm.di * dj, rhs = diag && ry >= 0 && ry < dj;
    if (ent || rhs) {
      const int e = ent ? (bm.off - base) + lane : csize + ry;
      if (ITEMS) {
        const double* p = part + (bm.it0 - cm.ibase) * kItemDoubles + (ent ? lane : 36 + ry);
        for (int q = 0; q < bm.nit; ++q) v -= p[q * kItemDoubles];
      } else if (bm.up1 > bm.up0) {
        const int wf = (bm.up0 - cm.ubase) / per, wl = (bm.up1 - 1 - cm.ubase) / per;
        for (int w = wf; w <= wl; ++w) v -= part[w * Et + e];
      }
      sm[e] = v;
    }
  }
  __syncthreads();
  // ---- diagonal block: wave 0 factors it in registers; then one thread per off-diagonal row solves
  //      x L_jj^T = v and stores to HBM
  double* Lw = C.Lval + base;
  if (dj == 6) chol_tail<6, NT>(sm, csize, part, Lw, C.y + cm.xoff, C.fail + g, tid);   // `part` is free again: reuse as L_jj staging
  else chol_tail<3, NT>(sm, csize, part, Lw, C.y + cm.xoff, C.fail + g, tid);
}

template <int NT>
__global__ __launch_bounds__(NT) void k_chol_level(BatchView V, CholView C, int lvl_begin) {
  extern __shared__ double sm[];
  chol_column<NT>(V, C, C.lcol[lvl_begin + blockIdx.x], sm);
}

// One step of the tail kernel: a *supernode* = s <= kMaxSn consecutive tail columns that form a chain of the elimination
// tree with nested structure (column p+1 is the parent of column p and owns exactly its remaining rows; blocks 3 or 6 wide).
//   1. external updates (sources outside the supernode) of all s columns at once: the item scheme of chol_column,
//   2. A(:,j) + lambda I - item partials -> an LDS panel holding the s columns back to back,
//   3. for p = 0 .. s-1: factor column p in the panel (diagonal block, row solves, forward substitution), then apply
//      it to the later columns of the panel straight from LDS:  S(i, p') -= L(i, p) L(p', p)^T,  rhs(p') -= L(p', p) y_p.
//      Block b' of column p' is block b' + (p' - p) of column p (nested, sorted row lists).
// A chain of s columns costs one latency-bound update phase instead of s; singletons (s = 1) skip step 3's updates.
template <int NT>
__device__ __forceinline__ void chol_supernode(const BatchView& V, const CholView& C, const SnMeta sn, const int g, double* sm) {
  constexpr int NW = NT / 64;
  constexpr int kMaxRows = 192;                 // rows of the supernode's first column (LDS table; the host checks)
  __shared__ int t_drow[kMaxRows];              // dim of row i of column 0 (row p = column p itself, then the common rows)
  __shared__ int t_rpre[kMaxRows + 1];          // prefix sums of t_drow
  __shared__ int t_col[kMaxSn][4];              // per column: Lval base, x offset, first block record, (unused)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int s = sn.s;
  const ColMeta cm0 = C.col[C.tail_cols[sn.q0]];
  const int nb0 = cm0.nb;                               // column p of the chain has nb0 - p blocks (nested row lists)
  auto poff = [&](int p) {
    int v = sn.poff[0];
#pragma unroll
    for (int q = 1; q <= kMaxSn; ++q) v = p == q ? sn.poff[q] : v;
    return v;
  };
  const int panel = poff(s);
  const double lambda = V.lm[g].lambda;
  const double* __restrict__ H = V.Hpp_diag;
  const double* __restrict__ L = C.Lval;
  const double* __restrict__ Y = C.y;
  double* part = sm + panel;
  const int ry = lane - 40;
  // table of the step (consumed after the barrier that ends phase 1)
  if (tid < nb0) {   // block i of column 0 starts at (sum of the row dims before it) x dim(column 0)
    const BlkMeta b0m = C.blk[cm0.b0 + tid];
    t_drow[tid] = b0m.di;
    t_rpre[tid] = (b0m.off - cm0.base) / cm0.dim;
    if (tid == 0) t_rpre[nb0] = cm0.csize / cm0.dim;
  }
  if (tid >= 64 && tid < 64 + s) {
    const ColMeta cp = C.col[C.tail_cols[sn.q0 + (tid - 64)]];
    t_col[tid - 64][0] = cp.base; t_col[tid - 64][1] = cp.xoff; t_col[tid - 64][2] = cp.b0;
  }
  // ---- 1. external updates (item scheme of chol_column; the target's width rides in the item record)
  {
    const int tr = (lane >> 1) & 1, tc = lane & 1;
    const int icount = sn.icount, chunk = sn.chunk;
    for (int it0 = wave * 16; it0 < icount; it0 += NT / 4) {
      const int it = it0 + (lane >> 2);
      const bool have = it < icount;
      const ItemMeta im = C.item[sn.ibase + min(it, icount - 1)];
      const int tpk = im.first.pk;
      const int di = (tpk & kUpdDi6) ? 6 : 3, dj = (tpk & kUpdDj6) ? 6 : 3;
      const bool diag = tpk & kUpdDiag;
      const bool tile = have && 3 * tr < di && 3 * tc < dj;
      const int tre = 3 * tr < di ? tr : 0, tce = 3 * tc < dj ? tc : 0;
      double acc[9], accy[3];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) accy[q] = 0;
      UpdMeta um = im.first;
      for (int k = 0; k < chunk; ++k) {
        const bool live = k < im.n;
        const UpdMeta nx = C.upd[im.u0 + min(k + 1, im.n - 1)];
        const int dk = (um.pk & kUpdDk6) ? 6 : 3;
        const double* A = L + um.ua + 3 * tre * dk;
        const double* B = L + um.ub + 3 * tce * dk;
        const double* yk = Y + um.ux;
        // K in halves of 3 (one half when the source column is 3 wide): 21 live doubles instead of 42 -- the 1024-thread
        // workgroup caps a lane at 128 VGPRs and the full tile spilled
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool on = live && 3 * h < dk;
          const int ko = 3 * h < dk ? 3 * h : 0;
          double a[9], bb[9], yv[3];
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int q = 0; q < 3; ++q) { a[rr * 3 + q] = A[rr * dk + ko + q]; bb[rr * 3 + q] = B[rr * dk + ko + q]; }
#pragma unroll
          for (int q = 0; q < 3; ++q) yv[q] = yk[ko + q];
          if (on) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) acc[rr * 3 + cc] += a[rr * 3] * bb[cc * 3] + a[rr * 3 + 1] * bb[cc * 3 + 1] + a[rr * 3 + 2] * bb[cc * 3 + 2];
              accy[rr] += a[rr * 3] * yv[0] + a[rr * 3 + 1] * yv[1] + a[rr * 3 + 2] * yv[2];
            }
          }
        }
        um = nx;
      }
      if (tile) {
        double* o = part + it * kItemDoubles;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] = acc[rr * 3 + cc];
          if (diag && tc == 0) o[36 + 3 * tr + rr] = accy[rr];
        }
      }
    }
  }
  __syncthreads();
  // ---- 2. A(:,j) + lambda I - partials -> panel; one wave per (column, block) pair, four pairs in flight per wave
  {
    int total = 0;
    for (int p = 0; p < s; ++p) total += nb0 - p;
    constexpr int kPipe = 4;
    for (int idx0 = wave; idx0 < total; idx0 += NW * kPipe) {
      BlkMeta bm[kPipe];
      int pp[kPipe], bb2[kPipe];
#pragma unroll
      for (int k = 0; k < kPipe; ++k) {
        const int idx = min(idx0 + k * NW, total - 1);
        int p = 0, bi = idx;
        while (bi >= nb0 - p) { bi -= nb0 - p; ++p; }
        pp[k] = p; bb2[k] = bi;
        bm[k] = C.blk[__builtin_amdgcn_readfirstlane(t_col[p][2] + bi)];
      }
      double hv[kPipe];
#pragma unroll
      for (int k = 0; k < kPipe; ++k) {
        const int p = pp[k], bi = bb2[k];
        const int dj = t_drow[p];
        const int r = lane / dj, c = lane - r * dj;
        const bool diag = (bi == 0);
        const bool ent = lane < bm[k].di * dj, rhs = diag && ry >= 0 && ry < dj;
        double v = 0;
        if (ent) {
          if (bm[k].src >= 0) v = bm[k].fmt ? H[bm[k].src + c * bm[k].di + r] : H[bm[k].src + r * dj + c];
          if (diag && r == c) v += lambda;
        } else if (rhs) {
          v = V.bvec[t_col[p][1] + ry];
        }
        hv[k] = v;
      }
#pragma unroll
      for (int k = 0; k < kPipe; ++k) {
        if (idx0 + k * NW >= total) continue;
        const int p = pp[k], bi = bb2[k];
        const int dj = t_drow[p];
        const bool diag = (bi == 0);
        const bool ent = lane < bm[k].di * dj, rhs = diag && ry >= 0 && ry < dj;
        if (ent || rhs) {
          double v = hv[k];
          const double* q0 = part + (bm[k].it0 - sn.ibase) * kItemDoubles + (ent ? lane : 36 + ry);
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;   // four independent chains (the LDS reads pipeline); fixed combination order
          int q = 0;
          for (; q + 4 <= bm[k].nit; q += 4) {
            s0 += q0[q * kItemDoubles]; s1 += q0[(q + 1) * kItemDoubles]; s2 += q0[(q + 2) * kItemDoubles]; s3 += q0[(q + 3) * kItemDoubles];
          }
          for (; q < bm[k].nit; ++q) s0 += q0[q * kItemDoubles];
          v -= (s0 + s1) + (s2 + s3);
          const int csz = (t_rpre[nb0] - t_rpre[p]) * dj;
          sm[poff(p) + (ent ? (t_rpre[p + bi] - t_rpre[p]) * dj + lane : csz + ry)] = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- 3. factor the panel column by column; offsets come from the row table (no global metadata on this path)
  for (int p = 0; p < s; ++p) {
    const int dp = t_drow[p];
    const int csz = (t_rpre[nb0] - t_rpre[p]) * dp;
    double* col = sm + poff(p);
    double* Lw = C.Lval + t_col[p][0];
    double* yo = C.y + t_col[p][1];
    if (s > 1) {
      if (dp == 6) chol_tail<6, NT, true>(col, csz, part, Lw, yo, C.fail + g, tid);
      else chol_tail<3, NT, true>(col, csz, part, Lw, yo, C.fail + g, tid);
    } else {
      if (dp == 6) chol_tail<6, NT>(col, csz, part, Lw, yo, C.fail + g, tid);
      else chol_tail<3, NT>(col, csz, part, Lw, yo, C.fail + g, tid);
    }
    if (p + 1 < s) {
      __syncthreads();
      int total = 0;
      for (int p2 = p + 1; p2 < s; ++p2) total += nb0 - p2;
      for (int idx = wave; idx < total; idx += NW) {
        int p2 = p + 1, b2 = idx;
        while (b2 >= nb0 - p2) { b2 -= nb0 - p2; ++p2; }
        const int d2 = t_drow[p2], di = t_drow[p2 + b2];
        const double* Ab = col + (t_rpre[p2 + b2] - t_rpre[p]) * dp;     // L(row p2 + b2, p)
        const double* Bb = col + (t_rpre[p2] - t_rpre[p]) * dp;          // L(p2, p)
        double* tgt = sm + poff(p2);
        if (lane < di * d2) {
          const int r = lane / d2, c = lane - r * d2;
          const double* pa = Ab + r * dp;
          const double* pb = Bb + c * dp;
          double d = pa[0] * pb[0] + pa[1] * pb[1] + pa[2] * pb[2];
          if (dp == 6) d += pa[3] * pb[3] + pa[4] * pb[4] + pa[5] * pb[5];
          tgt[(t_rpre[p2 + b2] - t_rpre[p2]) * d2 + lane] -= d;
        } else if (b2 == 0 && ry >= 0 && ry < d2) {
          const double* pb = Bb + ry * dp;
          const double* yp = col + csz;                                  // y_p sits behind the column's entries
          double d = pb[0] * yp[0] + pb[1] * yp[1] + pb[2] * yp[2];
          if (dp == 6) d += pb[3] * yp[3] + pb[4] * yp[4] + pb[5] * yp[5];
          tgt[(t_rpre[nb0] - t_rpre[p2]) * d2 + ry] -= d;
        }
      }
      __syncthreads();
    }
  }
}

// Top of the elimination tree: once a graph's levels are at most a couple of columns wide, a launch per level
// only buys launch latency and cold caches.  One workgroup per graph walks those columns in elimination order,
// a supernode at a time; the barrier between steps orders the L / y stores of one step before the loads of the next
// (same CU).
template <int NT>
__global__ __launch_bounds__(NT) void k_chol_tail(BatchView V, CholView C) {
  extern __shared__ double sm[];
  const int g = blockIdx.x;
  if (!V.lm[g].in_trial) return;
  const int n1 = C.sn_ptr[g + 1];
  for (int n = C.sn_ptr[g]; n < n1; ++n) {
    chol_supernode<NT>(V, C, C.sn[n], g, sm);
    __threadfence_block();
    __syncthreads();
  }
}

// forward substitution only (multi right-hand-side form, used for marginals): y_j = L_jj^-1 (b_j - sum_k L_jk y_k)
__global__ __launch_bounds__(64) void k_chol_forward_level(CholView C, int lvl_begin, const double* __restrict__ rhs, double* __restrict__ y) {
  __shared__ double t[8];
  const int j = C.lvl_cols[lvl_begin + blockIdx.x];
  const ColMeta cm = C.col[j];
  const size_t vo = (size_t)blockIdx.y * C.dim;
  const int lane = threadIdx.x;
  const int dj = cm.dim;
  const BlkMeta bm = C.blk[cm.b0];
  const double* __restrict__ L = C.Lval;
  if (lane < dj) {
    double a = rhs[vo + cm.xoff + lane];
    for (int u = bm.up0; u < bm.up1; ++u) {
      const UpdMeta um = C.upd[u];
      const int dk = (um.pk & kUpdDk6) ? 6 : 3;
      const double* pa = L + um.ua + lane * dk;
      const double* yk = y + vo + um.ux;
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
    for (int r = 0; r < dj; ++r) y[vo + cm.xoff + r] = t[r];
  }
}

// backward substitution of one column by a team of 8 * Q lanes:   x_j = L_jj^-T (y_j - sum_i L_ij^T x_i)
// lane (c, q): component c of block slice q (blocks 1 + q, 1 + q + Q, ...); the slices are summed with xor
// shuffles, the back-substitution runs across the 8 component lanes with shuffles; nothing goes through LDS.
// Q = 8: one column per wave (long columns, latency-bound levels); Q = 1: eight columns per wave (leaf levels).
template <int Q>
__device__ __forceinline__ void chol_backward_column(const CholView& C, const int j, const double* __restrict__ y, double* x,
                                                     const size_t vo, const int lt) {
  const int c = lt & 7, q = lt >> 3;
  const ColMeta cm = C.col[j];
  const int dj = cm.dim;
  const double* __restrict__ L = C.Lval;
  const int cc = min(c, dj - 1);   // idle lanes of the team shadow the last component
  const double* D = L + cm.base;
  double acc = 0;
  for (int bi = 1 + q; bi < cm.nb; bi += Q) {
    const BlkMeta bm = C.blk[cm.b0 + bi];
    const double* Bk = L + bm.off + cc;
    const double* xi = x + vo + bm.xoff_row;
    double s = Bk[0] * xi[0] + Bk[dj] * xi[1] + Bk[2 * dj] * xi[2];
    if (bm.di == 6) s += Bk[3 * dj] * xi[3] + Bk[4 * dj] * xi[4] + Bk[5 * dj] * xi[5];
    acc += s;
  }
  if (Q > 1) acc += __shfl_xor(acc, 8, 64);
  if (Q > 2) acc += __shfl_xor(acc, 16, 64);
  if (Q > 4) acc += __shfl_xor(acc, 32, 64);
  double t = y[vo + cm.xoff + cc] - acc;
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

// levels top-down, 64 / (8 Q) columns per wave; blockIdx.y = right-hand side
template <int Q>
__global__ __launch_bounds__(64) void k_chol_backward_level(CholView C, int lvl_begin, int n, const double* __restrict__ y, double* x,
                                                           const LmState* __restrict__ lm) {
  constexpr int kCols = 8 / Q;
  const int p = blockIdx.x * kCols + threadIdx.x / (8 * Q);
  const int j = C.lvl_cols[lvl_begin + min(p, n - 1)];
  const bool on = p < n && !(lm && !lm[C.col[j].graph].in_trial);
  if (on) chol_backward_column<Q>(C, j, y, x, (size_t)blockIdx.y * C.dim, threadIdx.x % (8 * Q));
}

// The top of the elimination tree (the columns k_chol_tail factors) backwards in one launch: one wave per
// graph walks its tail columns from the root down; a fence orders the x stores of a column before the loads
// of the next.
__global__ __launch_bounds__(64) void k_chol_backward_head(CholView C, const double* __restrict__ y, double* x, const LmState* __restrict__ lm) {
  const int g = blockIdx.x;
  if (lm && !lm[g].in_trial) return;
  const size_t vo = (size_t)blockIdx.y * C.dim;
  const int q0 = C.tail_ptr[g];
  for (int q = C.tail_ptr[g + 1] - 1; q >= q0; --q) {
    chol_backward_column<8>(C, C.tail_cols[q], y, x, vo, threadIdx.x);
    __threadfence_block();   // same wave, same CU: workgroup scope is enough (an agent-scope fence writes back L2)
  }
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
// host symbolic phase
// ------------------------------------------------------------------------------------------------
namespace {

struct GraphSym {
  std::vector<int> order;                  // elimination order (local node ids)
  std::vector<std::vector<int>> cstruct;   // per node: higher-ordered neighbours at elimination time
};

// minimum degree with explicit fill (the block graphs here have ~1e4 nodes and fill ~1.7x)
void min_degree(int n, std::vector<std::vector<int>>& adj, GraphSym& out) {
  std::vector<char> done(n, 0);
  using Item = std::pair<int, int>;  // (degree, node); lazy deletion
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  for (int v = 0; v < n; ++v) { std::sort(adj[v].begin(), adj[v].end()); pq.push({(int)adj[v].size(), v}); }
  out.order.clear(); out.order.reserve(n);
  out.cstruct.assign(n, {});
  std::vector<int> merged;
  while (!pq.empty()) {
    const Item it = pq.top(); pq.pop();
    const int v = it.second;
    if (done[v] || it.first != (int)adj[v].size()) continue;
    done[v] = 1;
    out.order.push_back(v);
    std::vector<int>& nb = adj[v];
    out.cstruct[v] = nb;
    for (int u : nb) {
      std::vector<int>& au = adj[u];
      // au <- (au U nb) \ {u, v}   (both sorted)
      merged.clear();
      merged.reserve(au.size() + nb.size());
      std::set_union(au.begin(), au.end(), nb.begin(), nb.end(), std::back_inserter(merged));
      au.clear();
      for (int w : merged) if (w != u && w != v) au.push_back(w);
      pq.push({(int)au.size(), u});
    }
    std::vector<int>().swap(adj[v]);
  }
}

template <typename T>
int up_to_dev(CholPlan& P, hipStream_t s, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  const size_t n = std::max<size_t>(h.size(), 1);
  SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T)));
  P.allocs.push_back(p);
  if (!h.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  *out = (const T*)p;
  return 0;
}

}  // namespace

int chol_plan_build(Batch& b) {
  if (b.chol) { chol_plan_free(b.chol); b.chol = nullptr; }
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  CholPlan* P = new CholPlan();
  b.chol = P;
  const BatchView& V = b.V;
  const int nPr = V.nPr, nLr = V.nLr, nrow = nPr + nLr;
  // global internal rows: pose row p -> p ; landmark row l -> nPr + l
  auto row_dim = [&](int r) { return r < nPr ? 6 : 3; };
  auto row_xoff = [&](int r) { return r < nPr ? 6 * r : 6 * nPr + 3 * (r - nPr); };
  // H block lookup: (min row, max row) -> (offset, stored as [min][max])
  std::unordered_map<uint64_t, int> hoff;
  hoff.reserve(b.ppoff.size() + b.plblk.size());
  auto key = [](int a, int c) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)c; };
  for (size_t i = 0; i < b.ppoff.size(); ++i) hoff[key(b.ppoff[i].first, b.ppoff[i].second)] = (int)(b.hpp_off_base + (int64_t)i * 36);
  for (size_t i = 0; i < b.plblk.size(); ++i) hoff[key(b.plblk[i].first, nPr + b.plblk[i].second)] = (int)(b.hpl_base + (int64_t)i * 18);
  // adjacency of the whole batch (graphs are disconnected components; ordered per graph)
  std::vector<std::vector<int>> adj(nrow);
  for (auto& pr : b.ppoff) { adj[pr.first].push_back(pr.second); adj[pr.second].push_back(pr.first); }
  for (auto& pr : b.plblk) { adj[pr.first].push_back(nPr + pr.second); adj[nPr + pr.second].push_back(pr.first); }

  std::vector<int> col_row;            // column -> internal row
  std::vector<int> row_col(nrow, -1);  // internal row -> column
  std::vector<std::vector<int>> cstruct_rows;  // per column: rows (internal ids) of off-diagonal blocks
  std::vector<int> col_graph;
  for (int g = 0; g < V.B; ++g) {
    const GraphSeg& sg = b.seg[g];
    const int n = sg.nprow + sg.nlrow;
    auto loc2row = [&](int v) { return v < sg.nprow ? sg.prow0 + v : nPr + sg.lrow0 + (v - sg.nprow); };
    auto row2loc = [&](int r) { return r < nPr ? r - sg.prow0 : sg.nprow + (r - nPr - sg.lrow0); };
    std::vector<std::vector<int>> ladj(n);
    for (int v = 0; v < n; ++v) {
      const int r = loc2row(v);
      ladj[v].reserve(adj[r].size());
      for (int w : adj[r]) ladj[v].push_back(row2loc(w));
    }
    GraphSym S;
    min_degree(n, ladj, S);
    const int c0 = (int)col_row.size();
    for (int s = 0; s < n; ++s) { const int r = loc2row(S.order[s]); row_col[r] = c0 + s; col_row.push_back(r); col_graph.push_back(g); }
    for (int s = 0; s < n; ++s) {
      std::vector<int> rows;
      for (int w : S.cstruct[S.order[s]]) rows.push_back(loc2row(w));
      cstruct_rows.push_back(std::move(rows));
    }
  }
  const int ncol = (int)col_row.size();
  // column block lists sorted by elimination position of the row
  std::vector<int> bp(ncol + 1, 0), boff, brow, bsrc, col_xoff(ncol), col_dim(ncol);
  std::vector<unsigned char> bfmt;
  std::vector<std::unordered_map<int, int>> colblk(ncol);  // per column: row column-id -> block id
  int64_t lnz = 0;
  int max_entries = 0;
  for (int j = 0; j < ncol; ++j) {
    const int rj = col_row[j];
    const int dj = row_dim(rj);
    col_xoff[j] = row_xoff(rj); col_dim[j] = dj;
    std::vector<int> rows_c;
    for (int r : cstruct_rows[j]) rows_c.push_back(row_col[r]);
    std::sort(rows_c.begin(), rows_c.end());
    bp[j] = (int)boff.size();
    // diagonal
    boff.push_back((int)lnz); brow.push_back(j);
    bsrc.push_back(rj < nPr ? rj * 36 : (int)(b.hll_base + (int64_t)(rj - nPr) * 9)); bfmt.push_back(0);
    colblk[j][j] = bp[j];
    int entries = dj * dj;
    lnz += dj * dj;
    for (int i : rows_c) {
      const int ri = col_row[i];
      const int di = row_dim(ri);
      colblk[j][i] = (int)boff.size();
      boff.push_back((int)lnz); brow.push_back(i);
      const int a = std::min(ri, rj), c = std::max(ri, rj);
      auto it = hoff.find(key(a, c));
      if (it == hoff.end()) { bsrc.push_back(-1); bfmt.push_back(0); }
      else { bsrc.push_back(it->second); bfmt.push_back(ri == a ? 0 : 1); }  // stored [min][max]; we need [i][j]
      lnz += di * dj;
      entries += di * dj;
    }
    max_entries = std::max(max_entries, entries);
    if (lnz >= ((int64_t)1 << 31)) { return set_error(SSLAM_ERR_INVALID, "Cholesky factor too large for int32 offsets"); }
  }
  bp[ncol] = (int)boff.size();
  const int nblk = (int)boff.size();
  // update lists: column k updates every (i, j) pair of its structure with pos(j) <= pos(i)
  std::vector<std::vector<std::array<int, 3>>> ulist(nblk);
  for (int k = 0; k < ncol; ++k) {
    const int k0 = bp[k] + 1, k1 = bp[k + 1];
    for (int p = k0; p < k1; ++p) {      // j = brow[p]
      const int j = brow[p];
      for (int q = p; q < k1; ++q) {     // i = brow[q], pos(i) >= pos(j)
        const int i = brow[q];
        auto it = colblk[j].find(i);
        if (it == colblk[j].end()) return set_error(SSLAM_ERR_NUMERIC, "symbolic factorisation inconsistent (missing fill block)");
        ulist[it->second].push_back({boff[q], boff[p], k});
      }
    }
  }
  std::vector<int> up(nblk + 1, 0), ua, ub, uk, ux;
  std::vector<unsigned char> udk;
  for (int t = 0; t < nblk; ++t) {
    for (auto& u : ulist[t]) {  // already ascending in k
      ua.push_back(u[0]); ub.push_back(u[1]); uk.push_back(u[2]);
      ux.push_back(col_xoff[u[2]]); udk.push_back((unsigned char)col_dim[u[2]]);
    }
    up[t + 1] = (int)ua.size();
  }
  // levels of the block elimination tree
  std::vector<int> level(ncol, 0);
  int nlev = 0;
  for (int j = 0; j < ncol; ++j) {
    if (bp[j + 1] - bp[j] > 1) { const int par = brow[bp[j] + 1]; level[par] = std::max(level[par], level[j] + 1); }
    nlev = std::max(nlev, level[j] + 1);
  }
  // tail of every graph: the levels from which on the graph is at most `tail_width` columns wide
  // (a batch keeps every level launch busy with other graphs' columns for longer, so its tails start lower)
  int tail_width = V.B >= 32 ? 6 : 2;
  if (const char* e = getenv("SSLAM_CHOL_TAIL_WIDTH")) tail_width = atoi(e);
  std::vector<char> is_tail(ncol, 0);
  std::vector<int> tail_ptr(V.B + 1, 0), tail_cols;
  {
    int c0 = 0;
    for (int g = 0; g < V.B; ++g) {
      int c1 = c0;
      while (c1 < ncol && col_graph[c1] == g) ++c1;
      int gl = 0;
      for (int j = c0; j < c1; ++j) gl = std::max(gl, level[j] + 1);
      std::vector<int> width(gl, 0);
      for (int j = c0; j < c1; ++j) width[level[j]]++;
      int cut = gl;
      while (cut > 0 && width[cut - 1] <= tail_width) --cut;
      // tail columns in post-order of their subtree (children before parents; any topological order is valid for the
      // sequential tail kernel): chains of the elimination tree become contiguous, which is what supernodes need
      for (int j = c0; j < c1; ++j)   // the tail kernel keeps a row table of 192 entries in LDS
        if (level[j] >= cut && bp[j + 1] - bp[j] > 192) { cut = gl; break; }
      std::vector<std::vector<int>> kids(c1 - c0);
      std::vector<int> roots;
      for (int j = c0; j < c1; ++j) {
        if (level[j] < cut) continue;
        is_tail[j] = 1;
        const int par = bp[j + 1] - bp[j] > 1 ? brow[bp[j] + 1] : -1;
        if (par >= 0) kids[par - c0].push_back(j); else roots.push_back(j);
      }
      std::vector<std::pair<int, size_t>> stack;   // (column, next child)
      for (int rt : roots) {
        stack.push_back({rt, 0});
        while (!stack.empty()) {
          auto& top = stack.back();
          const std::vector<int>& ch = kids[top.first - c0];
          if (top.second < ch.size()) { const int nx = ch[top.second++]; stack.push_back({nx, 0}); }
          else { tail_cols.push_back(top.first); stack.pop_back(); }
        }
      }
      tail_ptr[g + 1] = (int)tail_cols.size();
      c0 = c1;
    }
  }
  P->tail_total = (int)tail_cols.size();
  P->lvl_maxlist.assign(nlev, 0);
  for (int j = 0; j < ncol; ++j)
    if (!is_tail[j])
      for (int t = bp[j]; t < bp[j + 1]; ++t) P->lvl_maxlist[level[j]] = std::max(P->lvl_maxlist[level[j]], up[t + 1] - up[t]);
  P->lvl_ptr.assign(nlev + 1, 0);
  P->lvl_nfactor.assign(nlev, 0);
  for (int j = 0; j < ncol; ++j) { P->lvl_ptr[level[j] + 1]++; if (!is_tail[j]) P->lvl_nfactor[level[j]]++; }
  for (int l = 0; l < nlev; ++l) P->lvl_ptr[l + 1] += P->lvl_ptr[l];
  std::vector<int> lvl_cols(ncol), cursor(P->lvl_ptr.begin(), P->lvl_ptr.end() - 1);
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j < ncol; ++j) if ((int)is_tail[j] == pass) lvl_cols[cursor[level[j]]++] = j;

  if (getenv("SSLAM_CHOL_DUMP")) {
    for (int l = 0; l < nlev; ++l) {
      long nbs = 0, ups = 0; int mnb = 0;
      for (int q = P->lvl_ptr[l]; q < P->lvl_ptr[l + 1]; ++q) {
        const int j = lvl_cols[q];
        nbs += bp[j + 1] - bp[j]; mnb = std::max(mnb, bp[j + 1] - bp[j]);
        ups += up[bp[j + 1]] - up[bp[j]];
      }
      fprintf(stderr, "[chol-dump] level %d cols %d blocks %ld maxnb %d updates %ld maxlist %d\n", l, P->lvl_ptr[l + 1] - P->lvl_ptr[l], nbs, mnb, ups, P->lvl_maxlist[l]);
    }
  }
  CholView& C = P->C;
  C.ncol = ncol; C.nlevels = nlev; C.dim = 6 * nPr + 3 * nLr;
  P->max_col_entries = max_entries; P->lnz = lnz;
  std::vector<ColMeta> colm(ncol);
  std::vector<BlkMeta> blkm(nblk);
  std::vector<UpdMeta> updm(ua.size());
  std::vector<ItemMeta> itemm;
  for (int j = 0; j < ncol; ++j)
    for (int t = bp[j]; t < bp[j + 1]; ++t) {
      const int toff = boff[t] - boff[bp[j]];
      if (toff > kUpdToffMask) return set_error(SSLAM_ERR_UNSUPPORTED, "a factor column has more than 2^20 entries");
      const int tpk = toff | (col_dim[brow[t]] == 6 ? kUpdDi6 : 0) | (t == bp[j] ? kUpdDiag : 0) | (col_dim[j] == 6 ? kUpdDj6 : 0);
      for (int u = up[t]; u < up[t + 1]; ++u) updm[u] = UpdMeta{ua[u], ub[u], ux[u], tpk | (udk[u] == 6 ? kUpdDk6 : 0)};
    }
  std::vector<int> col_lds(ncol, 0), col_et(ncol, 0);   // LDS doubles the column needs with items; entries + rhs
  auto col_csize = [&](int j) { const int last = bp[j + 1] - 1; return boff[last] - boff[bp[j]] + col_dim[brow[last]] * col_dim[j]; };
  // items of one column; updates whose source column is flagged in `skip` (the column's own supernode) are left out
  auto build_items = [&](int j, int chunk, const std::vector<char>* skip) {
    const int ibase = (int)itemm.size();
    for (int t = bp[j]; t < bp[j + 1]; ++t) {
      const int it0 = (int)itemm.size();
      int u = up[t];
      while (u < up[t + 1]) {
        if (skip && (*skip)[uk[u]]) { ++u; continue; }
        int n = 1;
        while (n < chunk && u + n < up[t + 1] && !(skip && (*skip)[uk[u + n]])) ++n;
        itemm.push_back(ItemMeta{updm[u], u, n, 0, 0});
        u += n;
      }
      blkm[t] = BlkMeta{boff[t], col_dim[brow[t]], bsrc[t], (int)bfmt[t], up[t], up[t + 1], brow[t], col_xoff[brow[t]], it0, (int)itemm.size() - it0};
    }
    const int icount = (int)itemm.size() - ibase;
    const int U = up[bp[j + 1]] - up[bp[j]];
    colm[j] = ColMeta{col_xoff[j], col_dim[j], col_graph[j], bp[j], bp[j + 1] - bp[j], boff[bp[j]], col_csize(j), up[bp[j]], U, ibase, icount, chunk};
    col_et[j] = col_csize(j) + col_dim[j];
    col_lds[j] = col_et[j] + std::max(icount, 1) * kItemDoubles;
  };
  for (int j = 0; j < ncol; ++j) {
    if (is_tail[j]) continue;
    const int U = up[bp[j + 1]] - up[bp[j]];
    const size_t mark = itemm.size();
    for (int chunk = std::max(1, (U + kItemsPerColumn - 1) / kItemsPerColumn);; ++chunk) {   // one pass of the slots (see the tail)
      itemm.resize(mark);
      build_items(j, chunk, nullptr);
      if ((int)(itemm.size() - mark) <= kItemsPerColumn || chunk >= U) break;
    }
  }
  // supernodes of every tail: maximal chains (next tail column = parent with exactly one row less, all blocks 6 x 6)
  std::vector<SnMeta> snm;
  std::vector<int> sn_ptr(V.B + 1, 0);
  std::vector<char> in_sn(ncol, 0);
  int sn_max = kMaxSn;
  if (const char* e = getenv("SSLAM_CHOL_SUPERNODE")) sn_max = std::max(1, std::min(kMaxSn, atoi(e)));
  P->tail_maxEt = 0;
  for (int g = 0; g < V.B; ++g) {
    int q = tail_ptr[g];
    while (q < tail_ptr[g + 1]) {
      int s_len = 1;
      while (s_len < sn_max && q + s_len < tail_ptr[g + 1]) {
        const int jl = tail_cols[q + s_len - 1], jn = tail_cols[q + s_len];
        const int nbl = bp[jl + 1] - bp[jl], nbn = bp[jn + 1] - bp[jn];
        if (nbl < 2 || brow[bp[jl] + 1] != jn || nbn != nbl - 1) break;
        ++s_len;
      }
      for (;;) {   // shrink until the panel + item partials fit the LDS
        for (int p = 0; p < s_len; ++p) in_sn[tail_cols[q + p]] = 1;
        int Uext = 0, panel = 0;
        for (int p = 0; p < s_len; ++p) {
          const int j = tail_cols[q + p];
          for (int u = up[bp[j]]; u < up[bp[j + 1]]; ++u) if (!(s_len > 1 && in_sn[uk[u]])) ++Uext;
          panel += col_csize(j) + col_dim[j];
        }
        int chunk = std::max(1, (Uext + kItemsPerColumn - 1) / kItemsPerColumn);
        const size_t mark = itemm.size();
        const int ibase = (int)mark;
        int icount = 0;
        for (;; ++chunk) {   // every target's list is cut separately: raise chunk until the items fit ONE pass of the 256 slots
          itemm.resize(mark);
          for (int p = 0; p < s_len; ++p) build_items(tail_cols[q + p], chunk, s_len > 1 ? &in_sn : nullptr);
          icount = (int)itemm.size() - ibase;
          if (icount <= kItemsPerColumn || chunk >= Uext) break;
        }
        const int lds = panel + std::max(icount, 1) * kItemDoubles;
        for (int p = 0; p < s_len; ++p) in_sn[tail_cols[q + p]] = 0;
        if (s_len > 1 && (size_t)lds * sizeof(double) > 144 * 1024) { itemm.resize(mark); --s_len; continue; }
        SnMeta rec{q, s_len, ibase, icount, chunk, lds, {0}};
        for (int p = 0, acc = 0; p <= kMaxSn + 1; ++p) {
          rec.poff[p] = acc;
          if (p < s_len) acc += col_csize(tail_cols[q + p]) + col_dim[tail_cols[q + p]];
        }
        snm.push_back(rec);
        P->tail_maxEt = std::max(P->tail_maxEt, lds);
        break;
      }
      q += s_len;
    }
    sn_ptr[g + 1] = (int)snm.size();
    if (g == 0 && getenv("SSLAM_CHOL_DUMP")) {
      int hist[kMaxSn + 1] = {0}, n3 = 0, nrow3 = 0;
      for (int n = sn_ptr[g]; n < sn_ptr[g + 1]; ++n) hist[snm[n].s]++;
      for (int q2 = tail_ptr[g]; q2 < tail_ptr[g + 1]; ++q2) {
        const int j = tail_cols[q2];
        if (col_dim[j] != 6) ++n3;
        for (int t = bp[j]; t < bp[j + 1]; ++t) if (col_dim[brow[t]] != 6) { ++nrow3; break; }
      }
      fprintf(stderr, "[chol-dump] graph 0 tail: %d columns, %d supernodes; sizes", tail_ptr[g + 1] - tail_ptr[g], sn_ptr[g + 1] - sn_ptr[g]);
      for (int k = 1; k <= kMaxSn; ++k) fprintf(stderr, " %d:%d", k, hist[k]);
      fprintf(stderr, "; 3-dim columns %d, columns with a 3-dim row %d\n", n3, nrow3);
    }
  }
  P->lvl_maxEt.assign(nlev, 0);
  P->lvl_maxItemLds.assign(nlev, 0);
  for (int j = 0; j < ncol; ++j) {
    if (is_tail[j]) continue;
    P->lvl_maxEt[level[j]] = std::max(P->lvl_maxEt[level[j]], col_et[j]);
    P->lvl_maxItemLds[level[j]] = std::max(P->lvl_maxItemLds[level[j]], col_lds[j]);
  }
  int rc;
  if ((rc = up_to_dev(*P, b.stream, colm, &C.col))) return rc;
  if ((rc = up_to_dev(*P, b.stream, blkm, &C.blk))) return rc;
  if ((rc = up_to_dev(*P, b.stream, updm, &C.upd))) return rc;
  if ((rc = up_to_dev(*P, b.stream, itemm, &C.item))) return rc;
  if ((rc = up_to_dev(*P, b.stream, sn_ptr, &C.sn_ptr))) return rc;
  if ((rc = up_to_dev(*P, b.stream, snm, &C.sn))) return rc;
  if ((rc = up_to_dev(*P, b.stream, lvl_cols, &C.lvl_cols))) return rc;
  std::vector<ColMeta> lcolm(ncol);
  for (int q = 0; q < ncol; ++q) lcolm[q] = colm[lvl_cols[q]];
  if ((rc = up_to_dev(*P, b.stream, lcolm, &C.lcol))) return rc;
  if ((rc = up_to_dev(*P, b.stream, tail_ptr, &C.tail_ptr))) return rc;
  if ((rc = up_to_dev(*P, b.stream, tail_cols, &C.tail_cols))) return rc;
  void* p = nullptr;
  SSLAM_HIP_TRY(hipMalloc(&p, (lnz + 64) * sizeof(double))); P->allocs.push_back(p); C.Lval = (double*)p;  // +64: clamped-lane over-reads
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (lnz + 64) * sizeof(double), b.stream));
  SSLAM_HIP_TRY(hipMalloc(&p, (C.dim + 8) * sizeof(double))); P->allocs.push_back(p); C.y = (double*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(p, 0, (C.dim + 8) * sizeof(double), b.stream));
  SSLAM_HIP_TRY(hipMalloc(&p, std::max(V.B, 1) * sizeof(int))); P->allocs.push_back(p); C.fail = (int*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(C.fail, 0, std::max(V.B, 1) * sizeof(int), b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  return 0;
}

int64_t chol_plan_lnz(const Batch& b) { return b.chol ? b.chol->lnz : 0; }
int chol_plan_levels(const Batch& b) { return b.chol ? b.chol->C.nlevels : 0; }

int chol_factor_and_forward(Batch& b) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  ScopedTimer t(b, "factor");
  hipLaunchKernelGGL(k_chol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  // LDS per level: item scheme (16 waves): entries + rhs + item partials; flat scheme: Et entries + per-wave partial
  // columns (nparts * Et <= kPartDoubles, nparts <= NW) + NW wave scratch areas
  auto lds_items = [&](int doubles) { return (size_t)std::max(doubles, 64) * sizeof(double); };
  auto lds_et = [&](int et, int nw) {
    et = std::max(et, 1);
    const int nparts = std::max(1, std::min(nw, kPartDoubles / et));
    return (size_t)(et + nparts * et + nw * kScr) * sizeof(double);
  };
  auto lds_for = [&](int l, int nw) { return nw >= 16 ? lds_items(P.lvl_maxItemLds[l]) : lds_et(P.lvl_maxEt[l], nw); };
  size_t lds_max = lds_items(P.tail_maxEt);
  for (int l = 0; l < C.nlevels; ++l) lds_max = std::max(lds_max, std::max(lds_for(l, 16), lds_for(l, 4)));
  if (lds_max > 160 * 1024) return set_error(SSLAM_ERR_UNSUPPORTED, "a factor column needs %zu B of LDS (> 160 KiB)", lds_max);
  if (lds_max > 64 * 1024) {
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_tail<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
  }
  for (int l = 0; l < C.nlevels; ++l) {
    const int n = P.lvl_nfactor[l];
    if (n <= 0) continue;
    // narrow levels are latency-bound (long update lists -> 16 waves per column); wide levels have enough
    // columns in flight to hide latency and run 4 waves per column
    if (P.lvl_maxlist[l] >= 32 && n < 1024) hipLaunchKernelGGL(k_chol_level<1024>, dim3(n), dim3(1024), lds_for(l, 16), b.stream, b.V, C, P.lvl_ptr[l]);
    else if ((P.lvl_maxlist[l] <= 6 && n >= 2048) || n >= 24576) hipLaunchKernelGGL(k_chol_level<64>, dim3(n), dim3(64), lds_for(l, 1), b.stream, b.V, C, P.lvl_ptr[l]);
    else hipLaunchKernelGGL(k_chol_level<256>, dim3(n), dim3(256), lds_for(l, 4), b.stream, b.V, C, P.lvl_ptr[l]);
  }
  if (P.tail_total > 0) hipLaunchKernelGGL(k_chol_tail<1024>, dim3(b.V.B), dim3(1024), lds_items(P.tail_maxEt), b.stream, b.V, C);
  hipLaunchKernelGGL(k_chol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky factor launch: %s", hipGetErrorString(e));
  return 0;
}

int chol_backward(Batch& b) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  ScopedTimer t(b, "solve");
  if (P.tail_total > 0) hipLaunchKernelGGL(k_chol_backward_head, dim3(b.V.B, 1), dim3(64), 0, b.stream, C, (const double*)C.y, b.V.x, (const LmState*)b.V.lm);
  for (int l = C.nlevels - 1; l >= 0; --l) {
    const int n = P.lvl_nfactor[l];
    if (n <= 0) continue;
    if (n >= 4096) hipLaunchKernelGGL(k_chol_backward_level<1>, dim3((n + 7) / 8, 1), dim3(64), 0, b.stream, C, P.lvl_ptr[l], n, (const double*)C.y, b.V.x, (const LmState*)b.V.lm);
    else hipLaunchKernelGGL(k_chol_backward_level<8>, dim3(n, 1), dim3(64), 0, b.stream, C, P.lvl_ptr[l], n, (const double*)C.y, b.V.x, (const LmState*)b.V.lm);
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
    if (P.tail_total > 0) hipLaunchKernelGGL(k_chol_backward_head, dim3(b.V.B, nr), dim3(64), 0, b.stream, C, (const double*)P.d_multi_y, P.d_multi_x, (const LmState*)nullptr);
    for (int l = C.nlevels - 1; l >= 0; --l) {
      const int n = P.lvl_nfactor[l];
      if (n > 0) hipLaunchKernelGGL(k_chol_backward_level<1>, dim3((n + 7) / 8, nr), dim3(64), 0, b.stream, C, P.lvl_ptr[l], n, (const double*)P.d_multi_y, P.d_multi_x, (const LmState*)nullptr);
    }
    SSLAM_HIP_TRY(hipMemcpyAsync(x_host + (size_t)r0 * C.dim, P.d_multi_x, bytes, hipMemcpyDeviceToHost, b.stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  }
  return 0;
}

}  // namespace sslam
