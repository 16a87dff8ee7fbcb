// Window multifrontal block Cholesky for the LM normal equations on gfx950 (FP64), numeric phase (round 3).
//
// Replaces g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with "lm_var"
// (reference src/ps_graph_slam/graph_slam.cpp:27,67-73).  Symbolic phase and the design: wchol_plan.hpp.
//   k_wchol_factor<CLS>     one workgroup per SEGMENT of the elimination tree: the active submatrix of the segment (a window of <= W
//                           block rows) lives in registers, four lanes per 6 x 6 tile; per pivot column: assemble (H blocks, update
//                           matrices of child segments), pivot panel -> LDS, 6 x 6 Cholesky + row solves, rank-6 update of the window;
//                           the factor leaves as one contiguous panel per column, the forward substitution rides along
//   k_wchol_backward<CLS>   the same segments top-down for x = L^-T y, the window's x in LDS
// Every phase of a step is written once, as a function of (thread id, that thread's registers, the workgroup's LDS block), and run
// by an executor: on the GPU one thread each with a barrier behind every phase, on the host (sslam_debug_wchol_solve, CPU-only
// tests of the plan AND of the kernel logic) a loop over the thread ids per phase.  A phase never reads an LDS word that another
// thread writes in the same phase, so both executions are equivalent.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sslam.h"
#include "graph_engine.hpp"
#include "wchol_plan.hpp"

namespace sslam {

struct WView {
  const WStep* step; const WRow* row; const WChild* child; const WFin* fin; const WSeg* seg; const unsigned char* wmap;
  const double* H;      // the [H || b] buffer (block offsets of WRow::hsrc are relative to it)
  const double* bvec;
  double* Lval; double* Uval; double* y; double* x;
  int* fail;            // [B]
};

constexpr int kWMaxCh = 4;   // child update matrices absorbed per pass
constexpr int kWKS = 16;     // columns whose step / row records are staged in LDS at a time

template <int W>
struct WShared {
  double P[W][36];        // the pivot column: P[slot] = F(row in slot, pivot column), 6 x 6 row-major, zero padded
  double Hs[W][36];       // factor: the H blocks of the NEXT pivot column, same orientation, staged one column ahead by the row threads
                          // backward: the pivot column of L as stored (panel), staged one column ahead
  double Lp[W * 36 + 2];  // factor: the finished pivot column in its HBM form (panel), written out with coalesced 16-byte stores
  double rhs[W][6];       // the right-hand-side row of the window (factor) / x of the window's rows (backward)
  double part[W][6];      // backward: per row of the column, its contribution to the pivot's right-hand side
  double yc[6], bs[6];    // y of the pivot column; b of the next pivot column
  double Lc[21], linv[6]; // factor: L_cc (packed lower by rows) and its reciprocal pivots
  WStep stS[kWKS];        // step records of the staged columns
  WRow stR[kWKS * W];     // their row records
  int ch_uoff[kWMaxCh], ch_m[kWMaxCh];
  unsigned char inv[kWMaxCh][W + 2];   // per child of the pass: window slot -> local row of the child's update matrix (0xFF none)
};

template <int S>
struct WThread {
  double acc[S][9];       // S tiles of the window, this lane's 3 x 3 quarter of each
  int ta[S], tb[S];       // window slots (a >= b) of the tiles; a >= W: no such tile
  double hreg[6];         // prefetch: this thread's row of an H block (factor) / elements of the next L panel (backward, up to 6)
  double breg;
};

#define SSLAM_HD __host__ __device__ __forceinline__
// the per-tile bodies of a phase are unrolled (a tile's registers need compile-time indices); without a fence the scheduler overlaps
// the operand loads of all of a thread's tiles and the register count doubles
#if defined(__HIP_DEVICE_COMPILE__)
#define SSLAM_TILE_FENCE() __builtin_amdgcn_sched_barrier(0)
// everything derived from a tile's slots is invariant over the columns of a segment; hoisted out of the column loop those address
// computations cost ~100 VGPRs for the whole kernel.  Launder the slots once per phase instead.
#define SSLAM_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define SSLAM_TILE_FENCE() ((void)0)
#define SSLAM_OPAQUE(x) ((void)0)
#endif

// tile index T = a (a + 1) / 2 + b  ->  (a, b), a >= b
SSLAM_HD void wtile_ab(int T, int& a, int& b) {
  int q = (int)((sqrtf(8.0f * (float)T + 1.0f) - 1.0f) * 0.5f);
  while (q * (q + 1) / 2 > T) --q;
  while ((q + 1) * (q + 2) / 2 <= T) ++q;
  a = q; b = T - q * (q + 1) / 2;
}
SSLAM_HD bool wbit(unsigned long long m, int s) { return (m >> s) & 1ull; }
SSLAM_HD int wrank(unsigned long long live, int s) {
  const unsigned long long below = live & ((1ull << s) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(below);
#else
  return __builtin_popcountll(below);
#endif
}

// 6 x 6 (or 3 x 3 in the leading block of a padded 6 x 6) Cholesky of the lower triangle of A (row-major, ld 6), in place in 21
// registers: L packed by rows -> Lc, reciprocal pivots -> linv.  Right-looking: one rsqrt + one multiply between consecutive pivots.
SSLAM_HD bool wchol6(const double* A, double* Lc /* 21 */, double* linv /* 6 */) {
  double a[21];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) a[r * (r + 1) / 2 + c] = A[r * 6 + c];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double d = a[c * (c + 1) / 2 + c];
    if (!(d > 0)) { ok = false; d = 1.0; }
#if defined(__HIP_DEVICE_COMPILE__)
    const double id = rsqrt(d);
#else
    const double id = 1.0 / sqrt(d);
#endif
    linv[c] = id;
    a[c * (c + 1) / 2 + c] = d * id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) a[r * (r + 1) / 2 + c] *= id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r)
#pragma unroll
      for (int c2 = c + 1; c2 <= r; ++c2) a[r * (r + 1) / 2 + c2] -= a[r * (r + 1) / 2 + c] * a[c2 * (c2 + 1) / 2 + c];
  }
#pragma unroll
  for (int q = 0; q < 21; ++q) Lc[q] = a[q];
  return ok;
}

// The same factorisation left-looking, column by column, out of LDS (A with row stride lda; L packed by rows into Lc, also in LDS):
// six live registers instead of 21 + 21 -- in the MFMA form the single thread that runs it shares its register budget with the
// accumulator tiles of the whole wave.
SSLAM_HD bool wchol6_lds(const double* A, int lda, double* Lc, double* linv) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double a[6];
#pragma unroll
    for (int r = c; r < 6; ++r) a[r] = A[r * lda + c];
#pragma unroll
    for (int k = 0; k < c; ++k) {
      const double lck = Lc[c * (c + 1) / 2 + k];
#pragma unroll
      for (int r = c; r < 6; ++r) a[r] -= Lc[r * (r + 1) / 2 + k] * lck;
    }
    double d = a[c];
    if (!(d > 0)) { ok = false; d = 1.0; }
#if defined(__HIP_DEVICE_COMPILE__)
    const double id = rsqrt(d);
#else
    const double id = 1.0 / sqrt(d);
#endif
    linv[c] = id;
    Lc[c * (c + 1) / 2 + c] = d * id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) Lc[r * (r + 1) / 2 + c] = a[r] * id;
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// factorisation of one segment.  Ex: executor (phase runner); W / NT / S: window slots, threads, tile registers per thread
// Per column the critical path touches no HBM load: the step / row records of kWKS columns at a time are staged in LDS, and the H
// blocks and b of column s + 1 are fetched by the row threads while column s is being eliminated.
// ------------------------------------------------------------------------------------------------
// Memory instructions are what these kernels are short of (a wave-wide 8-byte access to 64 different places costs the L1 as much as 64
// cache lines): every global access below is a run of consecutive doubles per thread, 16 bytes at a time where the alignment allows,
// and the factor leaves through LDS as one contiguous stream per column.
struct alignas(16) WD2 { double a, b; };
// row thread t = (row of the column, STORED row of its H block): fetch that stored row (contiguous in HBM) and b
template <int W, int S>
SSLAM_HD void wchol_prefetch(int tid, WThread<S>& ts, const WShared<W>& sm, const WView& C, int sn /* index in the staged chunk */, int rbase) {
  const WStep st = sm.stS[sn];
  const int dj = (st.piv >> 8) & 255, nr = st.piv >> 16;
  const int ri = tid / 6, r = tid - 6 * ri;
#pragma unroll
  for (int q = 0; q < 6; ++q) ts.hreg[q] = 0.0;
  ts.breg = 0.0;
  if (ri >= nr) return;
  const WRow wr = sm.stR[st.row0 - rbase + ri];
  const int di = (wr.slot >> 8) & 255, fmt = (wr.slot >> 16) & 1;
  // the block is stored rows x cols = (fmt ? dj x di : di x dj), row-major: stored row r holds `len` consecutive doubles
  const int srows = fmt ? dj : di, len = fmt ? di : dj;
  if (wr.hsrc >= 0 && r < srows) {
    const double* Hr = C.H + wr.hsrc + r * len;
    if (len == 6) {                              // 48 bytes on a 16-byte boundary (6-wide blocks start at even offsets)
      const WD2* h2 = reinterpret_cast<const WD2*>(Hr);
      const WD2 v0 = h2[0], v1 = h2[1], v2 = h2[2];
      ts.hreg[0] = v0.a; ts.hreg[1] = v0.b; ts.hreg[2] = v1.a; ts.hreg[3] = v1.b; ts.hreg[4] = v2.a; ts.hreg[5] = v2.b;
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) ts.hreg[q] = Hr[q];
    }
  }
  if (ri == 0 && r < dj) ts.breg = C.bvec[st.xoff + r];
}
// ... and park what was fetched for column `sn` in LDS (Hs in panel orientation: rows of the slot x pivot columns; bs)
template <int W, int S>
SSLAM_HD void wchol_stage(int tid, WThread<S>& ts, WShared<W>& sm, int sn, int rbase) {
  const WStep st = sm.stS[sn];
  const int nr = st.piv >> 16;
  const int ri = tid / 6, r = tid - 6 * ri;
  if (ri >= nr) return;
  const WRow wr = sm.stR[st.row0 - rbase + ri];
  const int slot = wr.slot & 255, fmt = (wr.slot >> 16) & 1;
  if (!fmt) {
#pragma unroll
    for (int q = 0; q < 6; ++q) sm.Hs[slot][r * 6 + q] = ts.hreg[q];
  } else {                                       // stored transposed: this thread holds column r of the panel block
#pragma unroll
    for (int q = 0; q < 6; ++q) sm.Hs[slot][q * 6 + r] = ts.hreg[q];
  }
  if (ri == 0) sm.bs[r] = ts.breg;
}

template <int W, int NT, int S, class Ex>
SSLAM_HD void wchol_factor_segment(Ex& ex, WShared<W>& sm, const WView& C, const WSeg sg, const double lambda) {
  // ---- start: empty window
  ex.phase([&](int tid, WThread<S>& ts) {
#pragma unroll
    for (int k = 0; k < S; ++k) {
#pragma unroll
      for (int q = 0; q < 9; ++q) ts.acc[k][q] = 0.0;
      wtile_ab(k * (NT / 4) + (tid >> 2), ts.ta[k], ts.tb[k]);
    }
    for (int t = tid; t < W * 6; t += NT) sm.rhs[t / 6][t % 6] = 0.0;
  });
  int rbase = 0;   // first row record of the staged chunk
  for (int s = 0; s < sg.nsteps; ++s) {
    const int sn = s % kWKS;
    if (sn == 0) {
      // ---- the records of the next kWKS columns -> LDS (one round trip), then this column's H blocks (exposed once per chunk)
      const int nst = sg.nsteps - s < kWKS ? sg.nsteps - s : kWKS;
      rbase = C.step[sg.step0 + s].row0;
      const int rend = C.step[sg.step0 + s + nst - 1].row0 + (C.step[sg.step0 + s + nst - 1].piv >> 16);
      ex.phase([&](int tid, WThread<S>& ts) {
        for (int t = tid; t < nst; t += NT) sm.stS[t] = C.step[sg.step0 + s + t];
        for (int t = tid; t < rend - rbase; t += NT) sm.stR[t] = C.row[rbase + t];
      });
      ex.phase([&](int tid, WThread<S>& ts) {
        wchol_prefetch<W, S>(tid, ts, sm, C, 0, rbase);
        wchol_stage<W, S>(tid, ts, sm, 0, rbase);
        if (nst > 1) wchol_prefetch<W, S>(tid, ts, sm, C, 1, rbase);
      });
    }
    const WStep st = sm.stS[sn];
    const int c = st.piv & 255, dj = (st.piv >> 8) & 255, nr = st.piv >> 16;
    const unsigned long long mask = (unsigned long long)st.mask_lo | ((unsigned long long)st.mask_hi << 32);
    // ---- update matrices of child segments that join at this column, kWMaxCh per pass
    for (int c0 = 0; c0 < st.nchild; c0 += kWMaxCh) {
      const int nc = st.nchild - c0 < kWMaxCh ? st.nchild - c0 : kWMaxCh;
      ex.phase([&](int tid, WThread<S>& ts) {
        for (int t = tid; t < kWMaxCh * (W + 2); t += NT) sm.inv[t / (W + 2)][t % (W + 2)] = 0xFF;
      });
      ex.phase([&](int tid, WThread<S>& ts) {
        for (int ch = 0; ch < nc; ++ch) {
          const WChild wc = C.child[st.child0 + c0 + ch];
          if (tid == 0) { sm.ch_uoff[ch] = wc.uoff; sm.ch_m[ch] = wc.m; }
          for (int p = tid; p < wc.m; p += NT) sm.inv[ch][C.wmap[wc.map0 + p]] = (unsigned char)p;
        }
      });
      ex.phase([&](int tid, WThread<S>& ts) {
        int tq = tid; SSLAM_OPAQUE(tq); const int tr = (tq >> 1) & 1, tc = tq & 1;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          SSLAM_TILE_FENCE();
          int a = ts.ta[k], b = ts.tb[k];
          SSLAM_OPAQUE(a); SSLAM_OPAQUE(b);
          if (a >= W || !wbit(mask, a) || !wbit(mask, b)) continue;
          for (int ch = 0; ch < nc; ++ch) {
            const int pa = sm.inv[ch][a], pb = sm.inv[ch][b];
            if (pa == 0xFF || pb == 0xFF) continue;
            const int hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
            const double* U = C.Uval + sm.ch_uoff[ch] + 36 * (hi * (hi + 1) / 2 + lo);
            if (pa >= pb) {
#pragma unroll
              for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) ts.acc[k][rr * 3 + cc] += U[(3 * tr + rr) * 6 + 3 * tc + cc];
            } else {   // the child stores the tile of this row pair the other way round
#pragma unroll
              for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) ts.acc[k][rr * 3 + cc] += U[(3 * tc + cc) * 6 + 3 * tr + rr];
            }
          }
        }
        for (int t = tid; t < W * 6; t += NT) {   // right-hand-side parts, one thread per (slot, component)
          const int slot = t / 6, r = t % 6;
          if (!wbit(mask, slot)) continue;
          double v = sm.rhs[slot][r];
          for (int ch = 0; ch < nc; ++ch) {
            const int p = sm.inv[ch][slot];
            if (p != 0xFF) { const int m = sm.ch_m[ch]; v += C.Uval[sm.ch_uoff[ch] + 36 * (m * (m + 1) / 2) + 6 * p + r]; }
          }
          sm.rhs[slot][r] = v;
        }
      });
    }
    // ---- A. the pivot cross of the window leaves the registers for the LDS panel, the column's H blocks (staged) join it there
    ex.phase([&](int tid, WThread<S>& ts) {
      int tq = tid; SSLAM_OPAQUE(tq); const int tr = (tq >> 1) & 1, tc = tq & 1;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        SSLAM_TILE_FENCE();
        int a = ts.ta[k], b = ts.tb[k];
        SSLAM_OPAQUE(a); SSLAM_OPAQUE(b);
        if (a >= W) continue;
        const bool inb = b == c;
        if (!(a == c || inb)) continue;
        const int other = inb ? a : b;              // diagonal tile: other == c
        if (!wbit(mask, other)) continue;           // not a row of this column: structurally zero
        // this lane's quarter of the panel block P[other] = F(row of `other`, pivot): the tile itself when its column slot is the
        // pivot, its transpose when its row slot is
        const int ptr = inb ? tr : tc, ptc = inb ? tc : tr;
        const double* hq = &sm.Hs[other][(3 * ptr) * 6 + 3 * ptc];
        double v[9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) v[rr * 3 + cc] = (inb ? ts.acc[k][rr * 3 + cc] : ts.acc[k][cc * 3 + rr]) + hq[rr * 6 + cc];
        if (other == c && ptr == ptc) {             // damping; the padding of a 3-wide pivot becomes an identity block
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) { if (3 * ptr + rr < dj) v[rr * 4] += lambda; else v[rr * 4] = 1.0; }
        }
        double* o = &sm.P[other][(3 * ptr) * 6 + 3 * ptc];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) { o[rr * 6 + cc] = v[rr * 3 + cc]; ts.acc[k][rr * 3 + cc] = 0.0; }
      }
    });
    // ---- B1. L_cc = chol(S_cc) and y_c = L_cc^-1 rhs_c by one thread (a chain of six dependent pivots: nothing to share out)
    ex.phase([&](int tid, WThread<S>& ts) {
      if (tid != 0) return;
      if (!wchol6(&sm.P[c][0], sm.Lc, sm.linv)) C.fail[sg.graph] = 1;
      double y[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double w = sm.rhs[c][r] + sm.bs[r];
#pragma unroll
        for (int q = 0; q < r; ++q) w -= sm.Lc[r * (r + 1) / 2 + q] * y[q];
        y[r] = w * sm.linv[r];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) sm.yc[r] = r < dj ? y[r] : 0.0;
    });
    // ---- B2. one thread per row of the panel: x L_cc^T = v (L_cc streamed from LDS); the finished rows are collected in LDS
    ex.phase([&](int tid, WThread<S>& ts) {
      const int ri = tid / 6, r = tid - 6 * ri;
      if (ri >= nr) return;
      const WRow wr = sm.stR[st.row0 - rbase + ri];
      const int slot = wr.slot & 255, di = (wr.slot >> 8) & 255;
      if (r >= di) return;
      double* out = sm.Lp + wr.lofs + r * dj;
      if (ri == 0) {                              // the pivot's own rows: row r of L_cc
        for (int q = 0; q < dj; ++q) out[q] = q <= r ? sm.Lc[r * (r + 1) / 2 + q] : 0.0;
        return;
      }
      double* v = &sm.P[slot][r * 6];
      double xr[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        double w = v[q];
#pragma unroll
        for (int q2 = 0; q2 < q; ++q2) w -= xr[q2] * sm.Lc[q * (q + 1) / 2 + q2];
        xr[q] = w * sm.linv[q];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) { v[q] = xr[q]; if (q < dj) out[q] = xr[q]; }
    });
    // ---- C. window(a, b) -= L(a, c) L(b, c)^T for the rows a, b of the column; right-hand-side row; the pivot's slot is free again.
    //         Then the next column's H blocks (fetched one column ago) are parked in LDS and the fetch of the one after is issued.
    ex.phase([&](int tid, WThread<S>& ts) {
      {                                             // the column's panel -> HBM: consecutive lanes, consecutive 16 bytes
        const WRow last = sm.stR[st.row0 - rbase + nr - 1];
        const int lsz = last.lofs + ((last.slot >> 8) & 255) * dj;
        WD2* dst = reinterpret_cast<WD2*>(C.Lval + st.loff);
        const WD2* src = reinterpret_cast<const WD2*>(sm.Lp);
        for (int e = tid; 2 * e < lsz; e += NT) dst[e] = src[e];
        if (tid < dj) C.y[st.xoff + tid] = sm.yc[tid];
      }
      int tq = tid; SSLAM_OPAQUE(tq); const int tr = (tq >> 1) & 1, tc = tq & 1;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        SSLAM_TILE_FENCE();
        int a = ts.ta[k], b = ts.tb[k];
        SSLAM_OPAQUE(a); SSLAM_OPAQUE(b);
        if (a >= W || a == c || b == c || !wbit(mask, a) || !wbit(mask, b)) continue;
        const double* A = &sm.P[a][(3 * tr) * 6];
        const double* Bm = &sm.P[b][(3 * tc) * 6];
        double bb[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) bb[q] = Bm[q];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          double ar[6];
#pragma unroll
          for (int q = 0; q < 6; ++q) ar[q] = A[rr * 6 + q];
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            double w = ts.acc[k][rr * 3 + cc];
#pragma unroll
            for (int q = 0; q < 6; ++q) w -= ar[q] * bb[cc * 6 + q];
            ts.acc[k][rr * 3 + cc] = w;
          }
        }
      }
      {
        const int ri = tid / 6, r = tid - 6 * ri;
        if (ri < nr) {
          const int slot = sm.stR[st.row0 - rbase + ri].slot & 255;
          if (ri == 0) sm.rhs[slot][r] = 0.0;
          else {
            double w = sm.rhs[slot][r];
#pragma unroll
            for (int q = 0; q < 6; ++q) w -= sm.P[slot][r * 6 + q] * sm.yc[q];
            sm.rhs[slot][r] = w;
          }
        }
      }
      if (s + 1 < sg.nsteps && sn + 1 < kWKS) {
        wchol_stage<W, S>(tid, ts, sm, sn + 1, rbase);
        if (s + 2 < sg.nsteps && sn + 2 < kWKS) wchol_prefetch<W, S>(tid, ts, sm, C, sn + 2, rbase);
      }
    });
  }
  // ---- end: what is left in the window is the update matrix for the parent segment
  if (sg.m > 0) {
    unsigned long long live = 0;
    for (int p = 0; p < sg.m; ++p) live |= 1ull << (C.fin[sg.fin0 + p].slot & 255);
    ex.phase([&](int tid, WThread<S>& ts) {
      int tq = tid; SSLAM_OPAQUE(tq); const int tr = (tq >> 1) & 1, tc = tq & 1;
      double* U = C.Uval + sg.uoff;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        SSLAM_TILE_FENCE();
        int a = ts.ta[k], b = ts.tb[k];
        SSLAM_OPAQUE(a); SSLAM_OPAQUE(b);
        if (a >= W || !wbit(live, a) || !wbit(live, b)) continue;
        const int p = wrank(live, a), q = wrank(live, b);
        double* o = U + 36 * (p * (p + 1) / 2 + q) + (3 * tr) * 6 + 3 * tc;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[rr * 6 + cc] = ts.acc[k][rr * 3 + cc];
      }
      for (int t = tid; t < W * 6; t += NT) {
        const int slot = t / 6, r = t % 6;
        if (wbit(live, slot)) U[36 * (sg.m * (sg.m + 1) / 2) + 6 * wrank(live, slot) + r] = sm.rhs[slot][r];
      }
    });
  }
}

// ------------------------------------------------------------------------------------------------
// backward substitution of one segment: x_c = L_cc^-T (y_c - sum_i L_ic^T x_i), columns in reverse order.  Same staging: the
// records of kWKS columns at a time in LDS, the panel of column s - 1 fetched while column s is solved.
// ------------------------------------------------------------------------------------------------
template <int W, int NT, int S>
SSLAM_HD void wback_prefetch(int tid, WThread<S>& ts, const WView& C, const WStep st, int lsz) {
  const double* Lp = C.Lval + st.loff;
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int e = tid + k * NT; ts.hreg[k] = e < lsz ? Lp[e] : 0.0; }
  ts.breg = tid < 6 && tid < ((st.piv >> 8) & 255) ? C.y[st.xoff + tid] : 0.0;
}
template <int W, int NT, int S>
SSLAM_HD void wback_stage(int tid, WThread<S>& ts, WShared<W>& sm, int lsz) {
  double* Lp = &sm.Hs[0][0];
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int e = tid + k * NT; if (e < lsz) Lp[e] = ts.hreg[k]; }
  if (tid < 6) sm.yc[tid] = ts.breg;
}

template <int W, int NT, int S, class Ex>
SSLAM_HD void wchol_backward_segment(Ex& ex, WShared<W>& sm, const WView& C, const WSeg sg) {
  static_assert(6 * NT >= W * 36, "a panel is fetched with at most six elements per thread");
  // the rows above the segment (its update-matrix rows) are final: their x comes from HBM
  ex.phase([&](int tid, WThread<S>& ts) {
    for (int t = tid; t < sg.m * 6; t += NT) {
      const int p = t / 6, r = t - 6 * p;
      const WFin f = C.fin[sg.fin0 + p];
      const int slot = f.slot & 255, d = (f.slot >> 8) & 255;
      sm.rhs[slot][r] = r < d ? C.x[f.xoff + r] : 0.0;
    }
  });
  int rbase = 0, s0 = 0;   // first row record / first column of the staged chunk
  auto panel_doubles = [&](const WStep& st) {   // rows of the panel x width of the column (the row records are staged)
    const WRow last = sm.stR[st.row0 - rbase + (st.piv >> 16) - 1];
    return last.lofs + ((last.slot >> 8) & 255) * ((st.piv >> 8) & 255);
  };
  for (int s = sg.nsteps - 1; s >= 0; --s) {
    if (s == sg.nsteps - 1 || s < s0) {
      // ---- the records of the previous kWKS columns -> LDS, then this column's panel (exposed once per chunk)
      s0 = s + 1 - kWKS > 0 ? s + 1 - kWKS : 0;
      const int nst = s + 1 - s0;
      rbase = C.step[sg.step0 + s0].row0;
      const int rend = C.step[sg.step0 + s].row0 + (C.step[sg.step0 + s].piv >> 16);
      ex.phase([&](int tid, WThread<S>& ts) {
        for (int t = tid; t < nst; t += NT) sm.stS[t] = C.step[sg.step0 + s0 + t];
        for (int t = tid; t < rend - rbase; t += NT) sm.stR[t] = C.row[rbase + t];
      });
      ex.phase([&](int tid, WThread<S>& ts) {
        const WStep st = sm.stS[s - s0];
        const int lsz = panel_doubles(st);
        wback_prefetch<W, NT, S>(tid, ts, C, st, lsz);
        wback_stage<W, NT, S>(tid, ts, sm, lsz);
        if (s - 1 >= s0) { const WStep sp = sm.stS[s - 1 - s0]; wback_prefetch<W, NT, S>(tid, ts, C, sp, panel_doubles(sp)); }
      });
    }
    const WStep st = sm.stS[s - s0];
    const int c = st.piv & 255, dj = (st.piv >> 8) & 255, nr = st.piv >> 16;
    // ---- every row of the column: its block's contribution L_ic^T x_i, one thread per (row, component of the pivot); the last six
    //      threads invert the pivots meanwhile
    ex.phase([&](int tid, WThread<S>& ts) {
      for (int t = tid; t < 6 * (nr - 1); t += NT) {
        const int ri = 1 + t / 6, q = t % 6;
        double w = 0.0;
        if (q < dj) {
          const WRow wr = sm.stR[st.row0 - rbase + ri];
          const int slot = wr.slot & 255, di = (wr.slot >> 8) & 255;
          const double* Lb = &sm.Hs[0][0] + wr.lofs + q;
          for (int r = 0; r < di; ++r) w += Lb[r * dj] * sm.rhs[slot][r];
        }
        sm.part[ri][q] = w;
      }
      if (tid >= NT - 6) { const int q = tid - (NT - 6); sm.linv[q] = q < dj ? 1.0 / sm.Hs[0][q * dj + q] : 0.0; }
    });
    // ---- t = y_c - sum of the contributions, one thread per component
    ex.phase([&](int tid, WThread<S>& ts) {
      if (tid >= 6) return;
      double w = sm.yc[tid];
      for (int ri = 1; ri < nr; ++ri) w -= sm.part[ri][tid];
      sm.bs[tid] = w;
    });
    // ---- x_c = L_cc^-T t (six dependent steps: one thread)
    ex.phase([&](int tid, WThread<S>& ts) {
      if (tid != 0) return;
      const double* Ld = &sm.Hs[0][0];
      double x[6];
#pragma unroll
      for (int r = 5; r >= 0; --r) {
        x[r] = 0.0;
        if (r < dj) {
          double w = sm.bs[r];
#pragma unroll
          for (int s2 = 5; s2 > r; --s2) if (s2 < dj) w -= Ld[s2 * dj + r] * x[s2];
          x[r] = w * sm.linv[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) sm.rhs[c][r] = x[r];
    });
    // ---- x_c -> HBM (one lane per component); the next column's panel (fetched while this one was solved) -> LDS; fetch of the one after
    ex.phase([&](int tid, WThread<S>& ts) {
      if (tid < dj) C.x[st.xoff + tid] = sm.rhs[c][tid];
      if (s - 1 >= s0) {
        const WStep sp = sm.stS[s - 1 - s0];
        wback_stage<W, NT, S>(tid, ts, sm, panel_doubles(sp));
        if (s - 2 >= s0) { const WStep sq = sm.stS[s - 2 - s0]; wback_prefetch<W, NT, S>(tid, ts, C, sq, panel_doubles(sq)); }
      }
    });
  }
}

// ================================================================================================
// MFMA form of the factorisation (the default of the window plan).
//
// PMC of the VALU form above (profiles/r3_pmc_wchol_v2_batch128.txt): the LDS pipe is 60 % busy, the VALUs 37 % -- the rank-6
// update of the window,  F(a, b) -= L(a, c) L(b, c)^T  over 3 x 3 quarters, reads 36 operand doubles from LDS per 54 FMAs, and the
// LDS (256 B/clk per CU) is shared by four SIMDs.  But the update of the whole window is ONE dense product,
//     F (R x R)  -=  P (R x 8)  P^T ,     R = 6 W scalar rows, P = the pivot panel (6 columns, padded to K = 8),
// i.e. exactly what the matrix cores do with a single operand double per lane and instruction: v_mfma_f64_16x16x4_f64, 1024 FMAs
// per instruction (this is the "dense Schur block that forms a GEMM" of BASELINE.json's north_star; SQ_INSTS_VALU_MFMA_MOPS_F64 /
// SQ_VALU_MFMA_BUSY_CYCLES in profiles/ are the evidence).  So here the window lives in the MFMA accumulator layout:
//   * scalar row / column index of the window = 6 * slot + component; 16 x 16 tiles (I >= J) of the lower triangle, tile t owned by
//     wave t mod NW as its (t / NW)-th tile; element (row 16 I + (lane >> 4) + 4 reg, column 16 J + (lane & 15)) in acc[k][reg]
//     (cdna_hip_programming.md section 3: C/D map of the f64 MFMA); diagonal tiles hold both triangles;
//   * per pivot column: the pivot's 6 scalar columns leave the accumulators for the LDS panel P (ds_add onto the H blocks that the
//     row threads staged there one column ahead) and are zeroed; 6 x 6 Cholesky + one thread per scalar row for the solves; then
//     two MFMAs (K = 4 + 4) per active tile with operands A = -P[16 I + (lane & 15)][lane >> 4 (+ 4)], B = P[16 J + ...] straight
//     from LDS; P is double buffered so that the next column's H blocks can be parked while this column's operands are still read.
// ================================================================================================
typedef double wv4d __attribute__((ext_vector_type(4)));

template <int W, int NW>
struct WMCfg {
  static constexpr int NTR = (6 * W + 15) / 16, ROWS = 16 * NTR, NTILES = NTR * (NTR + 1) / 2, TPW = (NTILES + NW - 1) / NW, NT = 64 * NW;
  static_assert(6 * W <= 64 * NW, "one thread per scalar row of the window");
};
constexpr int kWPS = 9;      // doubles per panel row in LDS (8 used): a stride of 18 banks keeps the 16 rows of an operand load apart

template <int W, int NW>
struct WSharedM {
  double P[1][WMCfg<W, NW>::ROWS][kWPS];   // pivot panel [row][k], k < 8 (columns 6, 7 stay zero); the next column's H blocks are parked after the update has read its operands
  double rhs[WMCfg<W, NW>::ROWS];          // right-hand-side row of the window, per scalar row
  double yc[8], bs[6];
  double Lc[21], linv[6];
  WStep stS[kWKS];
  WRow stR[kWKS * W];
  int slot_tab[W];                         // per slot: (column number + 1) << 8 | row index in that column (tagged: no reset needed)
  int ch_uoff[kWMaxCh], ch_m[kWMaxCh];
  unsigned char inv[kWMaxCh][W + 2];
};
template <int TPW>
struct WThreadM {
  double acc[TPW][4];     // this lane's 4 elements of each of its wave's tiles
  double hreg[6];         // prefetch: this thread's scalar row of the next pivot column's H block
  double breg;
  int rnext, rcur;        // this thread's scalar row in the next / current pivot column: lofs | di << 24 | pivot << 28 | valid << 29, or 0
  int tI[TPW], tJ[TPW];   // tile row / column of the wave's k-th tile (compile-time constants with one wave; wave-uniform otherwise); -1: none
};
#if defined(__HIP_DEVICE_COMPILE__)
#define SSLAM_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define SSLAM_UNIFORM(x) (x)
#endif

// the thread of scalar row i looks its row up in column `sn` of the staged chunk (slot_tab tagged with that column) and fetches it
template <int W, int NW, int TPW>
SSLAM_HD void wm_prefetch(int tid, WThreadM<TPW>& ts, const WSharedM<W, NW>& sm, const WView& C, int sn, int s_abs, int rbase) {
#pragma unroll
  for (int q = 0; q < 6; ++q) ts.hreg[q] = 0.0;
  ts.breg = 0.0; ts.rnext = 0;
  if (tid >= 6 * W) return;
  const int slot = tid / 6, r = tid - 6 * slot;
  const int e = sm.slot_tab[slot];
  if ((e >> 8) != s_abs + 1) return;           // this slot holds no row of that column
  const WStep st = sm.stS[sn];
  const int dj = (st.piv >> 8) & 255, ri = e & 255;
  const WRow wr = sm.stR[st.row0 - rbase + ri];
  const int di = (wr.slot >> 8) & 255, fmt = (wr.slot >> 16) & 1;
  ts.rnext = wr.lofs | (di << 24) | ((ri == 0 ? 1 : 0) << 28) | (1 << 29);
  if (wr.hsrc >= 0 && r < di) {
    const double* Hb = C.H + wr.hsrc;
#pragma unroll
    for (int q = 0; q < 6; ++q) if (q < dj) ts.hreg[q] = fmt ? Hb[q * di + r] : Hb[r * dj + q];
  }
  if (ri == 0 && r < dj) ts.breg = C.bvec[st.xoff + r];
}
// ... and writes its whole row of the panel of that column: H (+ damping / identity padding on the pivot's diagonal), zeros elsewhere
template <int W, int NW, int TPW>
SSLAM_HD void wm_stage(int tid, WThreadM<TPW>& ts, WSharedM<W, NW>& sm, int sn, int s_abs, double lambda) {
  ts.rcur = ts.rnext;
  if (tid >= WMCfg<W, NW>::ROWS) return;
  double* o = &sm.P[0][tid][0];
  const bool piv = (ts.rcur >> 28) & 1;
  const int r = tid % 6;
  const int dj = (sm.stS[sn].piv >> 8) & 255;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    double v = q < 6 ? ts.hreg[q] : 0.0;
    if (piv && q == r) v = r < dj ? v + lambda : 1.0;
    o[q] = v;
  }
  if (piv) sm.bs[r] = ts.breg;
}

template <int W, int NW, class Ex>
SSLAM_HD void wchol_factor_segment_mfma(Ex& ex, WSharedM<W, NW>& sm, const WView& C, const WSeg sg, const double lambda) {
  using Cfg = WMCfg<W, NW>;
  constexpr int TPW = Cfg::TPW, NT = Cfg::NT, ROWS = Cfg::ROWS, NTILES = Cfg::NTILES;
  ex.phase([&](int tid, WThreadM<TPW>& ts) {
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ts.acc[k][r] = 0.0;
      const int t = k * NW + (NW == 1 ? 0 : SSLAM_UNIFORM(tid >> 6));
      int I = -1, J = -1;
      if (t < NTILES) wtile_ab(t, I, J);
      ts.tI[k] = NW == 1 ? I : SSLAM_UNIFORM(I); ts.tJ[k] = NW == 1 ? J : SSLAM_UNIFORM(J);
    }
    ts.rnext = ts.rcur = 0;
    for (int t = tid; t < ROWS; t += NT) sm.rhs[t] = 0.0;
    for (int t = tid; t < W; t += NT) sm.slot_tab[t] = 0;
  });
  int rbase = 0;
  for (int s = 0; s < sg.nsteps; ++s) {
    const int sn = s % kWKS;
    if (sn == 0) {
      // ---- the records of the next kWKS columns -> LDS; this column's rows are looked up and fetched at once (exposed per chunk)
      const int nst = sg.nsteps - s < kWKS ? sg.nsteps - s : kWKS;
      rbase = C.step[sg.step0 + s].row0;
      const int rend = C.step[sg.step0 + s + nst - 1].row0 + (C.step[sg.step0 + s + nst - 1].piv >> 16);
      ex.phase([&](int tid, WThreadM<TPW>& ts) {
        for (int t = tid; t < nst; t += NT) sm.stS[t] = C.step[sg.step0 + s + t];
        for (int t = tid; t < rend - rbase; t += NT) sm.stR[t] = C.row[rbase + t];
      });
      ex.phase([&](int tid, WThreadM<TPW>& ts) {      // slot table of this column
        const WStep st0 = sm.stS[0];
        const int nr0 = st0.piv >> 16;
        for (int ri = tid; ri < nr0; ri += NT) sm.slot_tab[sm.stR[st0.row0 - rbase + ri].slot & 255] = ((s + 1) << 8) | ri;
      });
      ex.phase([&](int tid, WThreadM<TPW>& ts) {
        wm_prefetch<W, NW, TPW>(tid, ts, sm, C, 0, s, rbase);
        wm_stage<W, NW, TPW>(tid, ts, sm, 0, s, lambda);
      });
      if (nst > 1) {
        ex.phase([&](int tid, WThreadM<TPW>& ts) {    // slot table of the next column, then its fetch
          const WStep st1 = sm.stS[1];
          const int nr1 = st1.piv >> 16;
          for (int ri = tid; ri < nr1; ri += NT) sm.slot_tab[sm.stR[st1.row0 - rbase + ri].slot & 255] = ((s + 2) << 8) | ri;
        });
        ex.phase([&](int tid, WThreadM<TPW>& ts) { wm_prefetch<W, NW, TPW>(tid, ts, sm, C, 1, s + 1, rbase); });
      }
    }
    const WStep st = sm.stS[sn];
    const int c = st.piv & 255, dj = (st.piv >> 8) & 255, nr = st.piv >> 16;
    const unsigned long long mask = (unsigned long long)st.mask_lo | ((unsigned long long)st.mask_hi << 32);
    const int pc0 = 6 * c;                       // first scalar column of the pivot
    double (*P)[kWPS] = sm.P[0];
    // ---- update matrices of child segments that join at this column (rare: once per child segment)
    for (int c0 = 0; c0 < st.nchild; c0 += kWMaxCh) {
      const int nc = st.nchild - c0 < kWMaxCh ? st.nchild - c0 : kWMaxCh;
      ex.phase([&](int tid, WThreadM<TPW>& ts) {
        for (int t = tid; t < kWMaxCh * (W + 2); t += NT) sm.inv[t / (W + 2)][t % (W + 2)] = 0xFF;
      });
      ex.phase([&](int tid, WThreadM<TPW>& ts) {
        for (int ch = 0; ch < nc; ++ch) {
          const WChild wc = C.child[st.child0 + c0 + ch];
          if (tid == 0) { sm.ch_uoff[ch] = wc.uoff; sm.ch_m[ch] = wc.m; }
          for (int p = tid; p < wc.m; p += NT) sm.inv[ch][C.wmap[wc.map0 + p]] = (unsigned char)p;
        }
      });
      ex.phase([&](int tid, WThreadM<TPW>& ts) {
        int l = tid & 63;
        SSLAM_OPAQUE(l);   // (lane-derived indices are invariant over the columns: hoisted, they cost a register each for the whole kernel)
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          SSLAM_TILE_FENCE();
          const int I = ts.tI[k], J = ts.tJ[k];
          if (I < 0) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + (l >> 4) + 4 * r, col = 16 * J + (l & 15);
            const int sr = row / 6, sc = col / 6;
            if (sr >= W || sc >= W || !wbit(mask, sr) || !wbit(mask, sc)) continue;
            const int rr = row - 6 * sr, rc = col - 6 * sc;
            double v = ts.acc[k][r];
            for (int ch = 0; ch < nc; ++ch) {
              const int pa = sm.inv[ch][sr], pb = sm.inv[ch][sc];
              if (pa == 0xFF || pb == 0xFF) continue;
              const double* U = C.Uval + sm.ch_uoff[ch];
              v += pa >= pb ? U[36 * (pa * (pa + 1) / 2 + pb) + rr * 6 + rc] : U[36 * (pb * (pb + 1) / 2 + pa) + rc * 6 + rr];
            }
            ts.acc[k][r] = v;
          }
        }
        if (tid < 6 * W) {                        // right-hand-side parts, the thread of the scalar row
          const int slot = tid / 6, r = tid - 6 * slot;
          if (wbit(mask, slot)) {
            double v = sm.rhs[tid];
            for (int ch = 0; ch < nc; ++ch) {
              const int p = sm.inv[ch][slot];
              if (p != 0xFF) { const int m = sm.ch_m[ch]; v += C.Uval[sm.ch_uoff[ch] + 36 * (m * (m + 1) / 2) + 6 * p + r]; }
            }
            sm.rhs[tid] = v;
          }
        }
      });
    }
    // ---- A. the pivot's scalar columns [pc0, pc0 + 6) leave the accumulators: added onto the staged H blocks in P, zeroed in the window
    ex.phase([&](int tid, WThreadM<TPW>& ts) {
      int l = tid & 63;
      SSLAM_OPAQUE(l);
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        SSLAM_TILE_FENCE();
        const int I = ts.tI[k], J = ts.tJ[k];
        if (I < 0) continue;
        // column strip: the tile's columns that are pivot columns, all its rows
        if (16 * J < pc0 + 6 && 16 * J + 16 > pc0) {
          const int q = 16 * J + (l & 15) - pc0;
          if (q >= 0 && q < 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * I + (l >> 4) + 4 * r;
#if defined(__HIP_DEVICE_COMPILE__)
              atomicAdd(&P[row][q], ts.acc[k][r]);      // ds_add_f64: one writer per entry (the H block is already there)
#else
              P[row][q] += ts.acc[k][r];
#endif
              ts.acc[k][r] = 0.0;
            }
          }
        }
        // row strip: the tile's rows that are pivot rows; columns of an earlier tile column are the transposed entries F(pivot, x)
        if (16 * I < pc0 + 6 && 16 * I + 16 > pc0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 16 * I + (l >> 4) + 4 * r - pc0;
            if (q >= 0 && q < 6) {
              if (I > J) {
                const int x = 16 * J + (l & 15);
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&P[x][q], ts.acc[k][r]);
#else
                P[x][q] += ts.acc[k][r];
#endif
              }
              ts.acc[k][r] = 0.0;                       // (diagonal tile: the mirror image of what the column strip took)
            }
          }
        }
      }
    });
    // ---- B1. L_cc = chol(S_cc), y_c; meanwhile the slot table of column s + 2 (fetched in phase C) is written
    ex.phase([&](int tid, WThreadM<TPW>& ts) {
      if (s + 2 < sg.nsteps && sn + 2 < kWKS) {
        const WStep s2 = sm.stS[sn + 2];
        const int nr2 = s2.piv >> 16;
        for (int ri = tid; ri < nr2; ri += NT) sm.slot_tab[sm.stR[s2.row0 - rbase + ri].slot & 255] = ((s + 3) << 8) | ri;
      }
      if (tid != 0) return;
      if (!wchol6_lds(&P[pc0][0], kWPS, sm.Lc, sm.linv)) C.fail[sg.graph] = 1;
      double y[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double w = sm.rhs[pc0 + r] + sm.bs[r];
#pragma unroll
        for (int q = 0; q < r; ++q) w -= sm.Lc[r * (r + 1) / 2 + q] * y[q];
        y[r] = w * sm.linv[r];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) { sm.yc[r] = r < dj ? y[r] : 0.0; if (r < dj) C.y[st.xoff + r] = y[r]; }
    });
    // ---- B2. the thread of every scalar row of the column: x L_cc^T = v, the factor leaves for HBM; right-hand-side row
    ex.phase([&](int tid, WThreadM<TPW>& ts) {
      if (tid >= 6 * W || !((ts.rcur >> 29) & 1)) return;
      const int r = tid % 6, di = (ts.rcur >> 24) & 15, lofs = ts.rcur & 0xFFFFFF;
      double* v = &P[tid][0];
      if ((ts.rcur >> 28) & 1) {                    // the pivot's own rows: row r of L_cc; out of the panel (the slot is retired)
        if (r < dj) { double* out = C.Lval + st.loff + r * dj; for (int q = 0; q < dj; ++q) out[q] = q <= r ? sm.Lc[r * (r + 1) / 2 + q] : 0.0; }
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = 0.0;
        sm.rhs[tid] = 0.0;
        return;
      }
      double xr[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        double w = v[q];
#pragma unroll
        for (int q2 = 0; q2 < q; ++q2) w -= xr[q2] * sm.Lc[q * (q + 1) / 2 + q2];
        xr[q] = w * sm.linv[q];
      }
      double w = sm.rhs[tid];
#pragma unroll
      for (int q = 0; q < 6; ++q) { v[q] = xr[q]; w -= xr[q] * sm.yc[q]; }
      sm.rhs[tid] = w;
      if (r < di) { double* out = C.Lval + st.loff + lofs + r * dj; for (int q = 0; q < dj; ++q) out[q] = xr[q]; }
    });
    // ---- C. window -= P P^T on the matrix cores; the next column's panel is parked, the fetch of the one after is issued
    {
      unsigned rowact = 0;                        // tile rows that hold a row of the column
      for (int q = 0; q < W; ++q) if (wbit(mask, q) && q != c) rowact |= (1u << ((6 * q) >> 4)) | (1u << ((6 * q + 5) >> 4));
      ex.mfma_update(&P[0][0], rowact, dj > 4 ? 2 : 1);
    }
    ex.phase([&](int tid, WThreadM<TPW>& ts) {
      if (s + 1 < sg.nsteps && sn + 1 < kWKS) {
        wm_stage<W, NW, TPW>(tid, ts, sm, sn + 1, s + 1, lambda);
        if (s + 2 < sg.nsteps && sn + 2 < kWKS) wm_prefetch<W, NW, TPW>(tid, ts, sm, C, sn + 2, s + 2, rbase);
      }
    });
  }
  // ---- end: what is left in the window is the update matrix for the parent segment (6 x 6 tiles over the live slots, p >= q)
  if (sg.m > 0) {
    unsigned long long live = 0;
    for (int p = 0; p < sg.m; ++p) live |= 1ull << (C.fin[sg.fin0 + p].slot & 255);
    ex.phase([&](int tid, WThreadM<TPW>& ts) {
      int l = tid & 63;
      SSLAM_OPAQUE(l);
      double* U = C.Uval + sg.uoff;
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        SSLAM_TILE_FENCE();
        const int I = ts.tI[k], J = ts.tJ[k];
        if (I < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + (l >> 4) + 4 * r, col = 16 * J + (l & 15);
          const int sr = row / 6, sc = col / 6;
          if (sr >= W || sc >= W || !wbit(live, sr) || !wbit(live, sc) || sr < sc) continue;
          const int rr = row - 6 * sr, rc = col - 6 * sc;
          const int p = wrank(live, sr), q = wrank(live, sc);
          double* o = U + 36 * (p * (p + 1) / 2 + q);
          o[rr * 6 + rc] = ts.acc[k][r];
          if (sr == sc) o[rc * 6 + rr] = ts.acc[k][r];   // a diagonal block that straddles two tile rows is stored on one side only
        }
      }
      if (tid < 6 * W) {
        const int slot = tid / 6, r = tid - 6 * slot;
        if (wbit(live, slot)) U[36 * (sg.m * (sg.m + 1) / 2) + 6 * wrank(live, slot) + r] = sm.rhs[tid];
      }
    });
  }
}

// ------------------------------------------------------------------------------------------------
// executors
// ------------------------------------------------------------------------------------------------
// Phase boundary on the GPU.  __syncthreads() is a workgroup-scope fence on ALL address spaces: the compiler drains every outstanding
// global load and store (s_waitcnt vmcnt(0)) in front of each barrier, i.e. every phase would wait for the L stores of the previous one
// and for the H prefetch that is meant to fly across the whole column.  The phases only communicate through LDS: one wave needs no
// barrier at all (its LDS operations execute in order), several waves need their LDS operations retired + s_barrier.
template <int S, int NT>
struct WGpuExec {
  WThread<S> ts;
  template <class F>
  __device__ __forceinline__ void phase(F&& f) {
    f((int)threadIdx.x, ts);
    if (NT <= 64) asm volatile("" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
};
template <int S>
struct WCpuExec {
  std::vector<WThread<S>> ts;
  int nt;
  explicit WCpuExec(int n) : ts(n), nt(n) {}
  template <class F>
  void phase(F&& f) { for (int t = 0; t < nt; ++t) f(t, ts[t]); }
};

// executors of the MFMA form: the phases as above, plus the window update -- on the GPU two v_mfma_f64_16x16x4_f64 per tile, on
// the host the same sums taken straight from the panel
template <int W, int NW>
struct WGpuExecM {
  using Cfg = WMCfg<W, NW>;
  WThreadM<Cfg::TPW> ts;
  template <class F>
  __device__ __forceinline__ void phase(F&& f) {
    f((int)threadIdx.x, ts);
    if (NW == 1) asm volatile("" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __device__ __forceinline__ void mfma_update(const double* P, unsigned rowact, int nk) {
#if defined(__HIP_DEVICE_COMPILE__)
    int l = threadIdx.x & 63;
    SSLAM_OPAQUE(l);
#pragma unroll
    for (int k = 0; k < Cfg::TPW; ++k) {
      SSLAM_TILE_FENCE();
      const int I = ts.tI[k], J = ts.tJ[k];
      if (I < 0 || !((rowact >> I) & 1u) || !((rowact >> J) & 1u)) continue;
      const double* pa = P + (16 * I + (l & 15)) * kWPS + (l >> 4);
      const double* pb = P + (16 * J + (l & 15)) * kWPS + (l >> 4);
      wv4d c = {ts.acc[k][0], ts.acc[k][1], ts.acc[k][2], ts.acc[k][3]};
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[0], pb[0], c, 0, 0, 0);
      if (nk > 1) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4], pb[4], c, 0, 0, 0);
      ts.acc[k][0] = c[0]; ts.acc[k][1] = c[1]; ts.acc[k][2] = c[2]; ts.acc[k][3] = c[3];
    }
    if (NW == 1) asm volatile("" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  }
};
template <int W, int NW>
struct WCpuExecM {
  using Cfg = WMCfg<W, NW>;
  std::vector<WThreadM<Cfg::TPW>> ts;
  WCpuExecM() : ts(Cfg::NT) {}
  template <class F>
  void phase(F&& f) { for (int t = 0; t < Cfg::NT; ++t) f(t, ts[t]); }
  void mfma_update(const double* P, unsigned rowact, int nk) {
    for (int tid = 0; tid < Cfg::NT; ++tid) {
      const int l = tid & 63;
      for (int k = 0; k < Cfg::TPW; ++k) {
        const int I = ts[tid].tI[k], J = ts[tid].tJ[k];
        if (I < 0 || !((rowact >> I) & 1u) || !((rowact >> J) & 1u)) continue;
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + (l >> 4) + 4 * r, col = 16 * J + (l & 15);
          double v = ts[tid].acc[k][r];
          for (int kk = 0; kk < 4 * nk; ++kk) v -= P[row * kWPS + kk] * P[col * kWPS + kk];
          ts[tid].acc[k][r] = v;
        }
      }
    }
  }
};

// minimum waves per SIMD asked of the compiler (register budget): the one-wave class at four waves per SIMD (<= 128 VGPRs)
constexpr int kWWaves[3] = {3, 2, 1};
template <int CLS>
__global__ __launch_bounds__(kWClass[CLS].nt, kWWaves[CLS]) void k_wchol_factor(WView C, const LmState* __restrict__ lm, int seg0) {
  constexpr int W = kWClass[CLS].wmax, NT = kWClass[CLS].nt, S = kWClass[CLS].S;
  __shared__ WShared<W> sm;
  const WSeg sg = C.seg[seg0 + blockIdx.x];
  if (!lm[sg.graph].in_trial) return;
  WGpuExec<S, NT> ex;
  wchol_factor_segment<W, NT, S>(ex, sm, C, sg, lm[sg.graph].lambda);
}
template <int CLS>
__global__ __launch_bounds__(kWClass[CLS].nt) void k_wchol_backward(WView C, const LmState* __restrict__ lm, int seg0) {
  constexpr int W = kWClass[CLS].wmax, NT = kWClass[CLS].nt, S = kWClass[CLS].S;
  __shared__ WShared<W> sm;
  const WSeg sg = C.seg[seg0 + blockIdx.x];
  if (lm && !lm[sg.graph].in_trial) return;
  WGpuExec<S, NT> ex;
  wchol_backward_segment<W, NT, S>(ex, sm, C, sg);
}
constexpr int kWMWaves[3] = {1, 8, 16};     // waves per workgroup of the MFMA form, by class
constexpr int kWMOcc[3] = {3, 2, 1};
template <int CLS>
__global__ __launch_bounds__(64 * kWMWaves[CLS]) __attribute__((amdgpu_waves_per_eu(kWMOcc[CLS], 8))) void k_wchol_factor_m(WView C, const LmState* __restrict__ lm, int seg0) {
  constexpr int W = kWClass[CLS].wmax, NW = kWMWaves[CLS];
  __shared__ WSharedM<W, NW> sm;
  const WSeg sg = C.seg[seg0 + blockIdx.x];
  if (!lm[sg.graph].in_trial) return;
  WGpuExecM<W, NW> ex;
  wchol_factor_segment_mfma<W, NW>(ex, sm, C, sg, lm[sg.graph].lambda);
}
__global__ void k_wchol_begin(BatchView V, WView C) {   // clear the failure flags of the graphs being solved
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) C.fail[g] = 0;
}
__global__ void k_wchol_end(BatchView V, WView C) {     // publish failures through the solver-agnostic flag
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) V.pcg_fail[g] = C.fail[g];
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
struct WPlan {
  WView C{};
  std::vector<int> launch_ptr, launch_cls;
  std::vector<void*> allocs;
  DevArena* arena = nullptr;
  int64_t lnz = 0, unz = 0;
  int nlevels = 0, nseg = 0;
};

void wchol_plan_free(WPlan* p) {
  if (!p) return;
  for (void* a : p->allocs) (void)hipFree(a);
  delete p;
}

namespace {
template <typename T>
int w_up(WPlan& P, hipStream_t s, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  const size_t n = h.size() + 4;
  if (P.arena) { p = P.arena->take(n * sizeof(T), true); if (!p) return set_error(SSLAM_ERR_HIP, "device allocation of %zu bytes failed", n * sizeof(T)); P.arena->note_direct(p, n * sizeof(T)); }
  else { SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T))); P.allocs.push_back(p); }
  if (!h.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  *out = (const T*)p;
  return 0;
}
int w_alloc(WPlan& P, void** q, size_t bytes) {
  if (P.arena) { *q = P.arena->take(bytes, true); if (*q) P.arena->note_direct(*q, bytes); return *q ? 0 : set_error(SSLAM_ERR_HIP, "device allocation of %zu bytes failed", bytes); }
  SSLAM_HIP_TRY(hipMalloc(q, bytes)); P.allocs.push_back(*q);
  return 0;
}
}  // namespace

int wchol_plan_build(Batch& b) {
  if (b.wchol) { wchol_plan_free(b.wchol); b.wchol = nullptr; }
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  SymIn in;
  chol_sym_input(b, in);
  WOpts opt;
  opt.from_env(b.V.B);
  WHost H;
  if (wchol_symbolic(in, opt, H)) return set_error(SSLAM_ERR_UNSUPPORTED, "window Cholesky plan: %s", H.error.c_str());
  WPlan* P = new WPlan();
  b.wchol = P;
  P->arena = b.arena;
  P->launch_ptr = H.launch_ptr; P->launch_cls = H.launch_cls; P->lnz = H.lnz; P->unz = H.unz; P->nlevels = H.nlevels; P->nseg = (int)H.seg.size();
  WView& C = P->C;
  int rc;
  if ((rc = w_up(*P, b.stream, H.step, &C.step))) return rc;
  if ((rc = w_up(*P, b.stream, H.row, &C.row))) return rc;
  if ((rc = w_up(*P, b.stream, H.child, &C.child))) return rc;
  if ((rc = w_up(*P, b.stream, H.fin, &C.fin))) return rc;
  if ((rc = w_up(*P, b.stream, H.seg, &C.seg))) return rc;
  if ((rc = w_up(*P, b.stream, H.wmap, &C.wmap))) return rc;
  void* p = nullptr;
  if ((rc = w_alloc(*P, &p, (H.lnz + 64) * sizeof(double)))) return rc;
  C.Lval = (double*)p;
  if ((rc = w_alloc(*P, &p, (H.unz + 64) * sizeof(double)))) return rc;
  C.Uval = (double*)p;
  if ((rc = w_alloc(*P, &p, ((size_t)H.dim + 8) * sizeof(double)))) return rc;
  C.y = (double*)p;
  if ((rc = w_alloc(*P, &p, std::max(b.V.B, 1) * sizeof(int)))) return rc;
  C.fail = (int*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(C.fail, 0, std::max(b.V.B, 1) * sizeof(int), b.stream));
  C.H = b.V.Hpp_diag; C.bvec = b.V.bvec; C.x = b.V.x;
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  return 0;
}

int64_t wchol_plan_lnz(const Batch& b) { return b.wchol ? b.wchol->lnz : 0; }
int64_t wchol_plan_unz(const Batch& b) { return b.wchol ? b.wchol->unz : 0; }
int wchol_plan_levels(const Batch& b) { return b.wchol ? b.wchol->nlevels : 0; }
int wchol_plan_launches(const Batch& b) { return b.wchol ? (int)b.wchol->launch_cls.size() : 0; }
int wchol_plan_segments(const Batch& b) { return b.wchol ? b.wchol->nseg : 0; }

int wchol_factor_and_forward(Batch& b) {
  WPlan& P = *b.wchol;
  const WView& C = P.C;
  ScopedTimer t(b, "factor");
  hipLaunchKernelGGL(k_wchol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  static const bool valu = [] { const char* e = getenv("SSLAM_WCHOL_VALU"); return e && atoi(e) != 0; }();   // the VALU form of the window update
  for (size_t l = 0; l < P.launch_cls.size(); ++l) {
    const int n = P.launch_ptr[l + 1] - P.launch_ptr[l];
    if (n <= 0) continue;
    if (!valu) {
      switch (P.launch_cls[l]) {
        case 0: hipLaunchKernelGGL(k_wchol_factor_m<0>, dim3(n), dim3(64 * kWMWaves[0]), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
        case 1: hipLaunchKernelGGL(k_wchol_factor_m<1>, dim3(n), dim3(64 * kWMWaves[1]), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
        default: hipLaunchKernelGGL(k_wchol_factor_m<2>, dim3(n), dim3(64 * kWMWaves[2]), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
      }
      continue;
    }
    switch (P.launch_cls[l]) {
      case 0: hipLaunchKernelGGL(k_wchol_factor<0>, dim3(n), dim3(kWClass[0].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
      case 1: hipLaunchKernelGGL(k_wchol_factor<1>, dim3(n), dim3(kWClass[1].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
      default: hipLaunchKernelGGL(k_wchol_factor<2>, dim3(n), dim3(kWClass[2].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
    }
  }
  hipLaunchKernelGGL(k_wchol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "window cholesky factor launch: %s", hipGetErrorString(e));
  return 0;
}

int wchol_backward(Batch& b) {
  WPlan& P = *b.wchol;
  const WView& C = P.C;
  ScopedTimer t(b, "solve");
  for (int l = (int)P.launch_cls.size() - 1; l >= 0; --l) {
    const int n = P.launch_ptr[l + 1] - P.launch_ptr[l];
    if (n <= 0) continue;
    switch (P.launch_cls[l]) {
      case 0: hipLaunchKernelGGL(k_wchol_backward<0>, dim3(n), dim3(kWClass[0].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
      case 1: hipLaunchKernelGGL(k_wchol_backward<1>, dim3(n), dim3(kWClass[1].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
      default: hipLaunchKernelGGL(k_wchol_backward<2>, dim3(n), dim3(kWClass[2].nt), 0, b.stream, C, (const LmState*)b.V.lm, P.launch_ptr[l]); break;
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "window cholesky solve launch: %s", hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// CPU execution of the same phases (plan + kernel logic tests on a box without a GPU): (H + lambda I) x = b for every graph
// ------------------------------------------------------------------------------------------------
template <int CLS>
static void wchol_emulate_segment(const WView& C, const WSeg sg, double lambda, bool backward) {
  constexpr int W = kWClass[CLS].wmax, NT = kWClass[CLS].nt, S = kWClass[CLS].S;
  const char* ev = getenv("SSLAM_WCHOL_VALU");
  const bool valu = ev && atoi(ev) != 0;
  if (!backward && !valu) {
    constexpr int NW = kWMWaves[CLS];
    static thread_local WSharedM<W, NW>* smm = nullptr;
    if (!smm) smm = new WSharedM<W, NW>();
    WCpuExecM<W, NW> ex;
    wchol_factor_segment_mfma<W, NW>(ex, *smm, C, sg, lambda);
    return;
  }
  static thread_local WShared<W>* sm = nullptr;
  if (!sm) sm = new WShared<W>();
  WCpuExec<S> ex(NT);
  if (backward) wchol_backward_segment<W, NT, S>(ex, *sm, C, sg);
  else wchol_factor_segment<W, NT, S>(ex, *sm, C, sg, lambda);
}

int wchol_emulate(const SymIn& in, const double* Hb, int64_t h_total, const double* lambda, double* x_out, int* fail_out, int64_t* stats /* [8] */) {
  WOpts opt;
  opt.from_env(in.B);
  WHost H;
  if (wchol_symbolic(in, opt, H)) return set_error(SSLAM_ERR_UNSUPPORTED, "window Cholesky plan: %s", H.error.c_str());
  std::vector<double> L((size_t)H.lnz + 8, NAN), U((size_t)H.unz + 8, NAN), y((size_t)H.dim + 8, NAN), x((size_t)H.dim + 8, NAN);
  std::vector<int> fail(std::max(in.B, 1), 0);
  WView C{};
  C.step = H.step.data(); C.row = H.row.data(); C.child = H.child.data(); C.fin = H.fin.data(); C.seg = H.seg.data(); C.wmap = H.wmap.data();
  const int64_t h_even = (h_total + 1) & ~(int64_t)1;
  C.H = Hb; C.bvec = Hb + h_even; C.Lval = L.data(); C.Uval = U.data(); C.y = y.data(); C.x = x.data(); C.fail = fail.data();
  auto run = [&](size_t l, bool backward) {
    for (int k = H.launch_ptr[l]; k < H.launch_ptr[l + 1]; ++k) {
      const WSeg sg = H.seg[k];
      switch (H.launch_cls[l]) {
        case 0: wchol_emulate_segment<0>(C, sg, lambda[sg.graph], backward); break;
        case 1: wchol_emulate_segment<1>(C, sg, lambda[sg.graph], backward); break;
        default: wchol_emulate_segment<2>(C, sg, lambda[sg.graph], backward); break;
      }
    }
  };
  for (size_t l = 0; l < H.launch_cls.size(); ++l) run(l, false);
  for (size_t l = H.launch_cls.size(); l-- > 0;) run(l, true);
  for (int i = 0; i < H.dim; ++i) x_out[i] = x[i];
  if (fail_out) for (int g = 0; g < in.B; ++g) fail_out[g] = fail[g];
  if (stats) {
    stats[0] = H.lnz; stats[1] = H.unz; stats[2] = (int64_t)H.seg.size(); stats[3] = (int64_t)H.launch_cls.size(); stats[4] = H.nlevels;
    stats[5] = (int64_t)H.row.size(); stats[6] = H.ncol;
    int64_t n2 = 0;
    for (const WSeg& s : H.seg) n2 += s.cls >= 1;
    stats[7] = n2;
  }
  return 0;
}

}  // namespace sslam
