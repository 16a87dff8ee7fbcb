// Factorisation of one piece through its FRONT tables (front_plan.hpp) -- included by sslam_chol.hip after its device helpers.
//
// Replaces, for the per-depth launches of large batches, chol_piece() and its record tables (reference
// src/ps_graph_slam/graph_slam.cpp:67-73,199-205 -> g2o BlockSolverX + LinearSolverCSparse).  Same arithmetic per block
// (tile_update_k, the 6 x 6 factor and the row solve of diag_factor / row_solve), same layout of L and y in HBM -- the backward
// substitution and the marginals read what this kernel writes.  What is different:
//   * ONE trip to HBM brings every table of the workgroup (a blob of a few hundred words); the row masks, the (column, local row) ->
//     L offset maps and the children's row maps are built in LDS from it while the H rows of the gather are in flight;
//   * the sources of a target tile are the set bits of  rows[i] & rows[j]  -- no update records; a child's update-matrix block is found
//     through its relative indices -- no assembly records; the update matrix is a dense lower triangle -- no item records;
//   * a level is two phases (two barriers, the record kernels take three): target tiles, then a TEAM of lanes per column -- every lane
//     factors the 6 x 6 diagonal block in registers itself (no one-lane diagonal phase: it was 16 % of a piece's clocks) and solves the
//     column's rows against it, lanes 0..5 keep row r of L_jj and write it after the barrier;
//   * mid and tail pieces (chains of columns) apply their internal updates right-looking, by source column, the pairs of a column's
//     blocks enumerated by arithmetic.
// Sums run in a fixed order (sources of dimension 6 ascending, then those of dimension 3, then the children in list order): bitwise repeatable.
#pragma once

namespace sslam {

struct FrontComp {
  int nc, NR, T, nch, ubase, usize, child0, tcum;
  unsigned long long p6;
  const uint2* bt;                 // own boundary table
  unsigned long long* rw;          // [NR] columns that hold local row r
  unsigned long long* cmask;       // [NR] children (the first 64) whose boundary holds local row r
  unsigned short* map;             // [nc][NR] local L offset of block (row r, column k)
  unsigned short* ycol;            // [nc] local y offset of column k
  unsigned char* trow;             // [T] boundary row << 1 | tile row
  unsigned char* inv;              // [nch][NR] index + 1 of local row r in child ch's boundary, 0: absent
};
__device__ __forceinline__ FrontComp front_comp(const unsigned* sB, unsigned char* sD, int q) {
  const uint4 W0 = *reinterpret_cast<const uint4*>(sB + kFrontHdr + kFrontComp * q), W1 = *reinterpret_cast<const uint4*>(sB + kFrontHdr + kFrontComp * q + 4);
  FrontComp c;
  c.nc = W0.x & 255; c.NR = c.nc + ((W0.x >> 8) & 255); c.T = (W0.x >> 16) & 255; c.nch = W0.x >> 24;
  c.ubase = (int)W0.y; c.usize = (int)W0.z; c.child0 = (int)(W1.x & 0xFFFF); c.tcum = (int)(W1.x >> 16);
  c.p6 = (unsigned long long)W1.y | ((unsigned long long)W1.z << 32);
  c.bt = reinterpret_cast<const uint2*>(sB + W0.w);
  unsigned char* d = sD + W1.w;
  c.rw = reinterpret_cast<unsigned long long*>(d); d += 8 * c.NR;
  c.cmask = reinterpret_cast<unsigned long long*>(d); d += 8 * c.NR;
  c.map = reinterpret_cast<unsigned short*>(d); d += front_pad8(2 * c.nc * c.NR);
  c.ycol = reinterpret_cast<unsigned short*>(d); d += front_pad8(2 * c.nc);
  c.trow = d; d += front_pad8(c.T);
  c.inv = d;
  return c;
}
// child header ch of the group: {Uval offset, doubles, boundary table}
struct FrontChild { int ubase, usize; const uint2* bt; };
__device__ __forceinline__ FrontChild front_child(const unsigned* sB, const unsigned* sChild, int ch) {
  const unsigned* h = sChild + kFrontChild * ch;
  return FrontChild{(int)h[0], (int)h[1], reinterpret_cast<const uint2*>(sB + h[3])};
}
__device__ __forceinline__ int front_child_block(const FrontChild& c, int qa, int qb, int di) {   // offset of block (qa, qb) of the child's update matrix
  const unsigned wb = c.bt[qb].x;
  return c.ubase + front_u_offset((int)(c.bt[qa].y & 0xFFFFFF), di, (wb >> 8) & 255, (wb >> 16) & 255);
}

// sum over the sources k of a target tile: the set bits of m -- first the columns of dimension 6, ascending, then those of dimension 3,
// ascending.  (One loop over all bits with the dimension tested per source made every wave that met both kinds in one iteration pay for
// both tile updates: the lanes of a wave walk different masks.  Two loops, each uniform in its arithmetic; the order of the sum is fixed
// either way.)
__device__ __forceinline__ void front_tile_sources(const FrontComp& cp, unsigned long long m, int li, int lj, int tr, int tc,
                                                   const double* __restrict__ smL, const double* __restrict__ smY, double (&acc)[9], double (&accy)[3]) {
  unsigned long long m6 = m & cp.p6, m3 = m & ~cp.p6;
  while (m6) {
    const int k = __builtin_ctzll(m6);
    m6 &= m6 - 1;
    tile_update_k<6>(smL, smY, cp.map[k * cp.NR + li], cp.map[k * cp.NR + lj], cp.ycol[k], tr, tc, acc, accy);
  }
  while (m3) {
    const int k = __builtin_ctzll(m3);
    m3 &= m3 - 1;
    tile_update_k<3>(smL, smY, cp.map[k * cp.NR + li], cp.map[k * cp.NR + lj], cp.ycol[k], tr, tc, acc, accy);
  }
}

// L_jj = chol(S_jj) in registers (lower triangle of a 6 x 6 scheme; a 3 x 3 block sits in its upper left corner, the rest an identity: one
// code path), t <- L_jj^-1 t, reciprocal pivots to inv: the arithmetic of diag_factor
__device__ __forceinline__ bool front_factor6(double (&a)[36], double (&tv)[6], double (&inv)[6]) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double d = a[c * 6 + c];
    if (!(d > 0)) { ok = false; d = 1.0; }
    const double id = rsqrt(d);
    inv[c] = id;
    a[c * 6 + c] = d * id;
    tv[c] *= id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) a[r * 6 + c] *= id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
#pragma unroll
      for (int c2 = c + 1; c2 <= r; ++c2) a[r * 6 + c2] -= a[r * 6 + c] * a[c2 * 6 + c];
      tv[r] -= a[r * 6 + c] * tv[c];
    }
  }
  return ok;
}

#define SSLAM_FSTAMP(k) if (dbg) { const long long now_ = clock64(); if (threadIdx.x == 0) dbg[k] += now_ - tprev; tprev = now_; }
template <int NT, bool RIGHT>
__device__ __forceinline__ void front_piece(const BatchView& V, const CholView& C, const PieceMeta pm, const FrontGrp fg, double* sm, long long* dbg = nullptr,
                                            const int* wait_p = nullptr, int wait_target = 0, int* wait_err = nullptr) {
  long long tprev = dbg ? clock64() : 0;
  const int tid = threadIdx.x;
  const int g = fg.graph;
  const double lambda = V.lm[g].lambda;      // (the graph's state travels with the blob: one trip, not two)
  const int in_trial = V.lm[g].in_trial;
  const int Lp = (pm.lsize + 1) & ~1, Yp = (pm.ysize + 1) & ~1;
  double* smL = sm;
  double* smY = smL + Lp;
  unsigned* sB = reinterpret_cast<unsigned*>(smY + Yp);
  unsigned char* sD = reinterpret_cast<unsigned char*>(sB + fg.words);
  // ---- 0. the blob -> LDS (one coalesced trip), the derived tables zeroed meanwhile
  {
    const uint4* src = reinterpret_cast<const uint4*>(C.fblob + fg.blob);
    uint4* dst = reinterpret_cast<uint4*>(sB);
    for (int q = tid; q < (fg.words >> 2); q += NT) dst[q] = src[q];
    unsigned long long* z = reinterpret_cast<unsigned long long*>(sD);
    for (int q = tid; q < (fg.dbytes >> 3); q += NT) z[q] = 0;
  }
  __syncthreads();
  if (!in_trial) return;
  SSLAM_FSTAMP(0)
  const unsigned h0 = sB[0], h1 = sB[1], h2 = sB[2];
  const int ncomp = h0 & 255, nlv = (h0 >> 8) & 255, nchild = h0 >> 16, nc = h1 & 0xFFFF, nb = h1 >> 16;
  const uint2* sMulti = reinterpret_cast<const uint2*>(sB + sB[8]);
  const uint4* sCol = reinterpret_cast<const uint4*>(sB + sB[3]);
  const uint4* sBlk = reinterpret_cast<const uint4*>(sB + sB[4]);
  const unsigned* sLv = sB + sB[5];
  const unsigned short* sTile = reinterpret_cast<const unsigned short*>(sB + sB[6]);
  const unsigned* sChild = sB + sB[7];
  const double* __restrict__ H = V.Hpp_diag;
  const double* __restrict__ U = C.Uval;
  // ---- 1. gather: A(:, piece) + lambda I and the rhs, minus the children's update-matrix blocks -> LDS.  One thread per (block, row),
  //         KG rows per thread in flight, a row's first child block (the plan names it) rides along with its H row.  The derived tables
  //         are built between the loads and their first use.
  auto derive = [&]() {
    for (int b = tid; b < nb; b += NT) {
      const uint4 bm = sBlk[b];
      const FrontComp cp = front_comp(sB, sD, bm.w & 255);
      const int kc = (bm.w >> 8) & 255, lr = (bm.y >> 16) & 63;
      cp.map[kc * cp.NR + lr] = (unsigned short)(bm.y & 0xFFFF);
      if (!(bm.y >> 31)) atomicOr(&cp.rw[lr], 1ull << kc);
    }
    for (int c = tid; c < nc; c += NT) {
      const uint4 col = sCol[c];
      const FrontComp cp = front_comp(sB, sD, col.y & 255);
      cp.ycol[(col.y >> 8) & 255] = (unsigned short)(col.x >> 16);
    }
    {   // own boundary tables (tile rows) and the copies of the children's (row maps): they follow one another, every entry tagged with its owner
      const unsigned w_bt0 = sB[kFrontHdr + 3], w_ch = sB[7];
      const uint2* e0 = reinterpret_cast<const uint2*>(sB + w_bt0);
      const int nown = (int)(w_ch - w_bt0) >> 1;
      for (int e = tid; e < nown; e += NT) {
        const uint2 w = e0[e];
        const FrontComp cp = front_comp(sB, sD, (w.x >> 25) | ((w.y >> 24) << 7));
        const int a = (int)(&e0[e] - cp.bt);
        const int t0 = 2 * ((w.x >> 8) & 255) + ((w.x >> 16) & 255);
        cp.trow[t0] = (unsigned char)(a << 1);
        if ((w.x >> 24) & 1) cp.trow[t0 + 1] = (unsigned char)((a << 1) | 1);
      }
      if (nchild > 0) {
        const unsigned w_cb0 = w_ch + kFrontChild * nchild;
        const uint2* c0 = reinterpret_cast<const uint2*>(sB + w_cb0);
        const int ncopy = (int)(sB[8] - w_cb0) >> 1;
        for (int e = tid; e < ncopy; e += NT) {
          const uint2 w = c0[e];
          const int ch = (w.x >> 25) | ((w.y >> 24) << 7);
          const unsigned* h = sChild + kFrontChild * ch;
          const int q = (int)(&c0[e] - reinterpret_cast<const uint2*>(sB + h[3]));
          if (q < 0 || q >= (int)(h[2] & 255)) continue;
          const FrontComp cp = front_comp(sB, sD, (h[2] >> 8) & 255);
          cp.inv[(h[2] >> 16) * cp.NR + (w.x & 255)] = (unsigned char)(q + 1);
          if ((h[2] >> 16) < 64) atomicOr(&cp.cmask[w.x & 255], 1ull << (h[2] >> 16));
        }
      }
    }
  };
  const int n1 = sB[9] & 0xFFFF, n2 = sB[9] >> 16, nmore = h2 >> 20;   // child sources by rank: the first ones are also named in the block records
  double m2[6], m2y = 0;
  {
    bool derived = false;
    if (wait_p) {
      // dependency-driven launches (k_chol_flow, k_chol_spec_round): the tables are in LDS and digested BEFORE the wait for the child pieces -- only
      // the gather itself (the children's update-matrix blocks) follows it
      derive();
      derived = true;
      __syncthreads();
      if (tid == 0) flow_wait(wait_p, wait_target, wait_err, C.fail + g);
      __syncthreads();
    }
    auto gather = [&](auto KGc, auto SRCc) {
      constexpr int KG = decltype(KGc)::value;
      constexpr bool SRC = decltype(SRCc)::value;
      const int nrow = nb * 6;
      for (int t0 = tid; t0 < nrow || !derived; t0 += NT * KG) {
        double v[KG][6], w[SRC ? KG : 1][6], rhsv[KG], uyv[SRC ? KG : 1];
#pragma unroll
        for (int g2 = 0; g2 < KG; ++g2) {
          const int t = min(t0 + g2 * NT, nrow - 1);
          const int b = t / 6, row = t - 6 * b;
          const uint4 bm = sBlk[b];
          const int di = ((bm.w >> 16) & 1) ? 6 : 3, dj = ((bm.w >> 17) & 1) ? 6 : 3;
          const bool fmt = (bm.y >> 30) & 1, dg = bm.y >> 31;
          const int rw = min(row, di - 1);
          const int src = (int)bm.x;
          const double* ph = H + max(src, 0) + (fmt ? rw : rw * dj);
          const int st = fmt ? di : 1;
          if (dj == 6 && !fmt) {
            const D2* ph2 = reinterpret_cast<const D2*>(ph);
#pragma unroll
            for (int c = 0; c < 3; ++c) { const D2 a2 = ph2[c]; v[g2][2 * c] = a2.a; v[g2][2 * c + 1] = a2.b; }
          } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[g2][c] = ph[min(c, dj - 1) * st];
          }
          rhsv[g2] = V.bvec[dg ? (int)sCol[(bm.y >> 22) & 255].z + rw : 0];
          if (SRC) {
            const bool on = (bm.z >> 20) != 0;
            const unsigned cw = sB[kFrontHdr + kFrontComp * (bm.w & 255) + 4];
            const FrontChild cc = front_child(sB, sChild, on ? (int)(cw & 0xFFFF) + (int)(bm.z & 255) : 0);
            const int qa = (bm.z >> 8) & 63, qb = (bm.z >> 14) & 63;
            const int uo = on ? front_child_block(cc, qa, qb, di) + rw * dj : 0;
            const double* pu = U + uo;
            if (dj == 6) {
              const D2* pu2 = reinterpret_cast<const D2*>(pu);
#pragma unroll
              for (int c = 0; c < 3; ++c) { const D2 b2 = pu2[c]; w[g2][2 * c] = b2.a; w[g2][2 * c + 1] = b2.b; }
            } else {
#pragma unroll
              for (int c = 0; c < 6; ++c) w[g2][c] = pu[min(c, dj - 1)];
            }
            uyv[g2] = U[(on && dg) ? cc.ubase + cc.usize + 6 * qa + rw : 0];
          }
        }
        if (t0 == tid) {
          // second child sources of the piece's blocks (a few per group; one (block, row) per thread): their loads travel with the main ones
          if (SRC && n2 > 0) {
            const int t = min(tid, min(n2, NT / 6) * 6 - 1);
            const uint2 me = sMulti[n1 + t / 6];
            const uint4 bm = sBlk[me.x & 0xFFFF];
            const int di = ((bm.w >> 16) & 1) ? 6 : 3, dj = ((bm.w >> 17) & 1) ? 6 : 3;
            const int rw = min(t % 6, di - 1);
            const FrontChild cc = front_child(sB, sChild, (int)(sB[kFrontHdr + kFrontComp * (bm.w & 255) + 4] & 0xFFFF) + (int)(me.x >> 16));
            const double* pu = U + front_child_block(cc, me.y & 255, (me.y >> 8) & 255, di) + rw * dj;
#pragma unroll
            for (int c = 0; c < 6; ++c) m2[c] = pu[min(c, dj - 1)];
            m2y = U[cc.ubase + cc.usize + 6 * (int)(me.y & 255) + rw];
          }
          if (!derived) { derive(); derived = true; }
        }
#pragma unroll
        for (int g2 = 0; g2 < KG; ++g2) {
          const int t = t0 + g2 * NT;
          if (t >= nrow) continue;
          const int b = t / 6, row = t - 6 * b;
          const uint4 bm = sBlk[b];
          const int di = ((bm.w >> 16) & 1) ? 6 : 3, dj = ((bm.w >> 17) & 1) ? 6 : 3;
          if (row >= di) continue;
          const bool diag = bm.y >> 31;
          const int src = (int)bm.x;
          const bool on = SRC && (bm.z >> 20) != 0;
          double vy = diag ? rhsv[g2] : 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) v[g2][c] = (src >= 0 ? v[g2][c] : 0.0) + ((diag && c == row) ? lambda : 0.0) - (on ? w[SRC ? g2 : 0][c] : 0.0);
          if (on && diag) vy -= uyv[SRC ? g2 : 0];
          double* o = smL + (bm.y & 0xFFFF) + row * dj;
#pragma unroll
          for (int c = 0; c < 6; ++c) if (c < dj) o[c] = v[g2][c];
          if (diag) smY[(sCol[(bm.y >> 22) & 255].x >> 16) + row] = vy;
        }
      }
    };
    if (nchild == 0) gather(std::integral_constant<int, 5>{}, std::false_type{});
    else gather(std::integral_constant<int, 3>{}, std::true_type{});
  }
  __syncthreads();
  SSLAM_FSTAMP(1)
  if (n2 + nmore > 0) {
    // further child sources, in list order: the second ones (loaded above) now that the first are in LDS, one (block, row) per thread; what does
    // not fit a pass, and third and later sources (rare), one source after the other by the first lanes of the workgroup
    auto apply = [&](const uint2 me, int row, const double (&u)[6], double uy) {
      const uint4 bm = sBlk[me.x & 0xFFFF];
      const int di = ((bm.w >> 16) & 1) ? 6 : 3, dj = ((bm.w >> 17) & 1) ? 6 : 3;
      if (row >= di) return;
      double* o = smL + (bm.y & 0xFFFF) + row * dj;
#pragma unroll
      for (int c = 0; c < 6; ++c) if (c < dj) o[c] -= u[c];
      if (bm.y >> 31) smY[(sCol[(bm.y >> 22) & 255].x >> 16) + row] -= uy;
    };
    const int npar = min(n2, NT / 6);
    if (tid < npar * 6) apply(sMulti[n1 + tid / 6], tid % 6, m2, m2y);
    __syncthreads();
    if (n2 > npar || nmore > 0) {
      // what is left, a chunk of NT / 6 sources at a time, one (source, row) per thread: the loads of a chunk travel together (ONE trip to HBM per
      // chunk -- the lists of the upper leaf depths and of the mid pieces ran to dozens of dependent trips by six lanes); a block has at most one
      // source of every rank and the list is sorted by rank, so the ranks of a chunk are applied one after the other: list order per block
      constexpr int CH = NT / 6;
      const int eend = n1 + n2 + nmore;
      for (int e0 = n1 + npar; e0 < eend; e0 += CH) {
        const int e = e0 + tid / 6, row = tid % 6;
        const bool have = tid < CH * 6 && e < eend;
        const uint2 me = sMulti[have ? e : e0];
        const uint4 bm = sBlk[me.x & 0xFFFF];
        const int di = ((bm.w >> 16) & 1) ? 6 : 3, dj = ((bm.w >> 17) & 1) ? 6 : 3;
        const int rw = min(row, di - 1);
        const FrontChild cc = front_child(sB, sChild, (int)(sB[kFrontHdr + kFrontComp * (bm.w & 255) + 4] & 0xFFFF) + (int)(me.x >> 16));
        const double* pu = U + front_child_block(cc, me.y & 255, (me.y >> 8) & 255, di) + rw * dj;
        double u[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) u[c] = pu[min(c, dj - 1)];
        const double uy = U[cc.ubase + cc.usize + 6 * (int)(me.y & 255) + rw];
        const int r0 = (int)(sMulti[e0].y >> 16), r1 = (int)(sMulti[min(e0 + CH, eend) - 1].y >> 16), myr = (int)(me.y >> 16);
        for (int r = r0; r <= r1; ++r) {
          if (have && myr == r) apply(me, row, u, uy);
          __syncthreads();
        }
      }
    }
  }
  SSLAM_FSTAMP(2)
  // ---- 2. the levels inside the piece, everything in LDS
  for (int il = 0; il < nlv; ++il) {
    const unsigned* lv = sLv + kFrontLv * il;
    const int t0 = (int)lv[1], t1 = (int)lv[2];
    // 2a. target tiles: one lane per 3 x 3 tile, its sources from the row masks (left-looking: the groups of leaf pieces)
    if (!RIGHT && t1 > t0) {
      for (int t = t0 + tid; t < t1; t += NT) {
        const unsigned e = sTile[t];
        const int tr = (e >> 1) & 1, tc = e & 1;
        const uint4 bm = sBlk[e >> 2];
        const FrontComp cp = front_comp(sB, sD, bm.w & 255);
        const int lr = (bm.y >> 16) & 63, kc = (bm.w >> 8) & 255;
        const int dj = ((bm.w >> 17) & 1) ? 6 : 3;
        double acc[9], accy[3];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) accy[q] = 0;
        front_tile_sources(cp, cp.rw[lr] & cp.rw[kc], lr, kc, tr, tc, smL, smY, acc, accy);
        double* o = smL + (bm.y & 0xFFFF);
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] -= acc[rr * 3 + cc];
        if ((bm.y >> 31) && tc == 0) {
          double* oy = smY + cp.ycol[kc];
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) oy[3 * tr + rr] -= accy[rr];
        }
      }
      __syncthreads();
      SSLAM_FSTAMP(3)
    }
    // 2b. a team of W lanes per column of the level (W: 8 .. NT, the widest that still gives every column its team; a lone column: the whole workgroup).  Every lane reads the
    //     diagonal block S_jj (lower triangle) and the rhs out of LDS and factors the block ITSELF in registers; lane r < 6 of the team owns
    //     row r of L_jj and y_r (y goes to LDS at once: only the team's first wave reads y, and its reads precede its writes); then the lanes solve the
    //     column's off-diagonal rows against the factor, x L_jj^T = v, W rows at a time.  Nothing but the rows of L_jj is written that another
    //     thread of this phase reads, and those are written AFTER the barrier; nobody but the final store reads them, so the next level's
    //     target tiles need not wait for them.
    const int c0 = lv[3] & 0xFFFF, ncl = (int)(lv[3] >> 16) - c0;
    int W = 8;
    while (W < NT && ncl * (2 * W) <= NT) W *= 2;   // (a lone column's team is the whole workgroup: its rows are solved in one round)
    const int teams = NT / W, team = tid / W, lt = tid % W;
    constexpr int kKeep = 2;   // (the plan refuses levels of more than kKeep * NT / 8 columns)
    double keep[kKeep][6];
#pragma unroll
    for (int r2 = 0; r2 < kKeep; ++r2) {
      const int cc = team + r2 * teams;
      if (cc < ncl) {
        const uint4 col = sCol[c0 + cc];
        const int dj = ((col.y >> 16) & 1) ? 6 : 3;
        const double* Sd = smL + (col.x & 0xFFFF);
        double* py = smY + (col.x >> 16);
        double a[36], tv[6], inv[6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) a[r * 6 + c] = (r < dj) ? Sd[r * dj + c] : (r == c ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < 6; ++c) tv[c] = (c < dj && lt < 64) ? py[c] : 0.0;   // (a team may span waves: only its first wave -- the one that writes y -- reads it)
        const bool ok = front_factor6(a, tv, inv);
        if (!ok && lt == 0) C.fail[g] = 1;
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (r == lt) {
#pragma unroll
            for (int c = 0; c < 6; ++c) keep[r2][c] = c <= r ? a[r * 6 + c] : 0.0;
            if (r < dj) py[r] = tv[r];
          }
        const int cb0 = col.w & 0x3FFF, nro = (int)(((col.w >> 14) & 255) - 1) * 6;
        for (int t = lt; t < nro; t += W) {
          const int b = cb0 + 1 + t / 6, row = t % 6;
          const uint4 bm = sBlk[b];
          const int di = ((bm.w >> 16) & 1) ? 6 : 3;
          if (row >= di) continue;
          double* pv = smL + (bm.y & 0xFFFF) + row * dj;
          double x[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            double w = c < dj ? pv[c] : 0.0;
#pragma unroll
            for (int s2 = 0; s2 < c; ++s2) w -= x[s2] * a[c * 6 + s2];
            x[c] = w * inv[c];
          }
#pragma unroll
          for (int c = 0; c < 6; ++c) if (c < dj) pv[c] = x[c];
        }
      }
    }
    __syncthreads();
    SSLAM_FSTAMP(4)
#pragma unroll
    for (int r2 = 0; r2 < kKeep; ++r2) {
      const int cc = team + r2 * teams;
      if (cc < ncl) {
        const uint4 col = sCol[c0 + cc];
        const int dj = ((col.y >> 16) & 1) ? 6 : 3;
        if (lt < dj) {
          double* pv = smL + (col.x & 0xFFFF) + lt * dj;
#pragma unroll
          for (int c = 0; c < 6; ++c) if (c < dj) pv[c] = keep[r2][c];
        }
      }
    }
    // 2c. right-looking (mid and tail pieces: chains of columns): a finished column updates every later block of its piece at once, all tile
    //     pairs in parallel, one tile update deep; one column per round (two columns of a level may meet in a target).  The pairs (p >= q) of the
    //     column's off-diagonal blocks with block q's row inside the piece are enumerated by arithmetic: 4 lanes per pair, one per 3 x 3 tile.
    if (RIGHT) {
      for (int c = c0; c < c0 + ncl; ++c) {
        const uint4 col = sCol[c];
        const int cb0 = col.w & 0x3FFF, m = (int)((col.w >> 14) & 255) - 1, mi = (int)(col.w >> 22);
        const int npair = mi * m - mi * (mi - 1) / 2;
        const bool dk6 = (col.y >> 16) & 1;
        const int yk = col.x >> 16;
        const FrontComp cp = front_comp(sB, sD, col.y & 255);
        for (int e = tid; e < 4 * npair; e += NT) {
          const int pr = e >> 2, tr = (e >> 1) & 1, tc = e & 1;
          // pairs are numbered q-major: q * m - q (q - 1) / 2 of them come before row q
          const float fm = (float)(2 * m + 1);
          int q = (int)((fm - sqrtf(fmaxf(fm * fm - 8.0f * (float)pr, 0.0f))) * 0.5f);
          q = max(0, min(q, mi - 1));
          while (q > 0 && q * m - q * (q - 1) / 2 > pr) --q;
          while (q + 1 < mi && (q + 1) * m - (q + 1) * q / 2 <= pr) ++q;
          const int p = q + (pr - (q * m - q * (q - 1) / 2));
          const uint4 ba = sBlk[cb0 + 1 + p], bb = sBlk[cb0 + 1 + q];
          const int di = ((ba.w >> 16) & 1) ? 6 : 3, dj = ((bb.w >> 16) & 1) ? 6 : 3;
          if (3 * tr >= di || 3 * tc >= dj || (p == q && tc > tr)) continue;
          const int li = (ba.y >> 16) & 63, lj = (bb.y >> 16) & 63;
          double acc[9], accy[3];
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[k] = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) accy[k] = 0;
          if (dk6) tile_update_k<6>(smL, smY, ba.y & 0xFFFF, bb.y & 0xFFFF, yk, tr, tc, acc, accy);
          else tile_update_k<3>(smL, smY, ba.y & 0xFFFF, bb.y & 0xFFFF, yk, tr, tc, acc, accy);
          double* o = smL + cp.map[lj * cp.NR + li];
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) o[(3 * tr + rr) * dj + 3 * tc + cc] -= acc[rr * 3 + cc];
          if (p == q && tc == 0) {
            double* oy = smY + cp.ycol[lj];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) oy[3 * tr + rr] -= accy[rr];
          }
        }
        __syncthreads();
      }
      SSLAM_FSTAMP(3)
    }
  }
  __syncthreads();   // (the last level's diagonal blocks are in LDS before the final store)
  // ---- 3. the update matrices: dense lower triangles over the boundary rows, one lane per 3 x 3 tile: own updates out of LDS + the
  //         children's blocks through their relative indices -> HBM
  {
    const unsigned cwl = sB[kFrontHdr + kFrontComp * (ncomp - 1) + 4], Tl = (sB[kFrontHdr + kFrontComp * (ncomp - 1)] >> 16) & 255;
    const int ntile = (int)(cwl >> 16) + (int)(Tl * (Tl + 1) / 2);
    for (int t = tid; t < ntile; t += NT) {
      int q = 0;
      while (q + 1 < ncomp && (int)(sB[kFrontHdr + kFrontComp * (q + 1) + 4] >> 16) <= t) ++q;
      const FrontComp cp = front_comp(sB, sD, q);
      const int u = t - cp.tcum;
      int p = (int)((sqrtf(8.0f * (float)u + 1.0f) - 1.0f) * 0.5f);
      while (p * (p + 1) / 2 > u) --p;
      while ((p + 1) * (p + 2) / 2 <= u) ++p;
      const int qq = u - p * (p + 1) / 2;
      const int ea = cp.trow[p], eb = cp.trow[qq];
      const int ia = ea >> 1, tra = ea & 1, ib = eb >> 1, trb = eb & 1;
      const int li = cp.nc + ia, lj = cp.nc + ib;
      const int di = ((cp.p6 >> li) & 1) ? 6 : 3, dj = ((cp.p6 >> lj) & 1) ? 6 : 3;
      const bool diag = ia == ib, wy = diag && trb == 0;
      // the first child block of the tile travels while the own updates are computed
      double c0v[9], c0y[3];
      int f = -1;
      {
        int qa = 0, qb = 0;
        // the children that hold both rows: the AND of two masks (a component with more than 64 children scans the rest)
        unsigned long long cm = cp.cmask[li] & cp.cmask[lj];
        if (cm) { f = __builtin_ctzll(cm); qa = cp.inv[f * cp.NR + li] - 1; qb = cp.inv[f * cp.NR + lj] - 1; }
        else
          for (int ch = 64; ch < cp.nch; ++ch) {
            const int a = cp.inv[ch * cp.NR + li], c2 = cp.inv[ch * cp.NR + lj];
            if (a && c2) { f = ch; qa = a - 1; qb = c2 - 1; break; }
          }
        const FrontChild cc = front_child(sB, sChild, min(cp.child0 + max(f, 0), max(nchild - 1, 0)));
        const double* o = U + (f >= 0 ? front_child_block(cc, qa, qb, di) : 0);
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int c = 0; c < 3; ++c) c0v[rr * 3 + c] = o[f >= 0 ? (3 * tra + rr) * dj + 3 * trb + c : 0];
        const double* oy = U + ((f >= 0 && wy) ? cc.ubase + cc.usize + 6 * qa + 3 * tra : 0);
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) c0y[rr] = oy[(f >= 0 && wy) ? rr : 0];
      }
      double acc[9], accy[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[k] = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) accy[k] = 0;
      front_tile_sources(cp, cp.rw[li] & cp.rw[lj], li, lj, tra, trb, smL, smY, acc, accy);
      if (f >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += c0v[k];
        if (wy) {
#pragma unroll
          for (int k = 0; k < 3; ++k) accy[k] += c0y[k];
        }
        unsigned long long cm = f < 63 ? (cp.cmask[li] & cp.cmask[lj]) >> (f + 1) << (f + 1) : 0ull;   // the children after f that hold both rows
        for (int ch = f + 1; ch < cp.nch; ++ch) {
          if (ch < 64) {
            if (!cm) { ch = 63; continue; }
            ch = __builtin_ctzll(cm);
            cm &= cm - 1;
          }
          const int a = cp.inv[ch * cp.NR + li], c2 = cp.inv[ch * cp.NR + lj];
          if (!(a && c2)) continue;
          const FrontChild cc = front_child(sB, sChild, cp.child0 + ch);
          const double* o = U + front_child_block(cc, a - 1, c2 - 1, di);
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[rr * 3 + c] += o[(3 * tra + rr) * dj + 3 * trb + c];
          if (wy) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) accy[rr] += U[cc.ubase + cc.usize + 6 * (a - 1) + 3 * tra + rr];
          }
        }
      }
      const unsigned wb = cp.bt[ib].x;
      double* o = C.Uval + cp.ubase + front_u_offset((int)(cp.bt[ia].y & 0xFFFFFF), di, (wb >> 8) & 255, (wb >> 16) & 255);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(3 * tra + rr) * dj + 3 * trb + c] = acc[rr * 3 + c];
      if (wy) {
        double* oy = C.Uval + cp.ubase + cp.usize + 6 * ia + 3 * tra;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) oy[rr] = accy[rr];
      }
    }
  }
  SSLAM_FSTAMP(5)
  // ---- 4. one coalesced stream out (the layouts of chol_piece)
  if (C.flat_L) {
    D2* dst = reinterpret_cast<D2*>(C.Lval + pm.lbase);
    const D2* src = reinterpret_cast<const D2*>(smL);
    for (int e = tid; e < (pm.lsize >> 1); e += NT) dst[e] = src[e];
  } else {
    double* dst = C.Lval + pm.lbase;
    const int nA = pm.n36, nB = pm.n18, nC = pm.nb - nA - nB, eA = 36 * nA, eB = eA + 18 * nB;
    const float rA = 1.0f / (float)max(nA, 1), rB = 1.0f / (float)max(nB, 1), rC = 1.0f / (float)max(nC, 1);
    for (int t = tid; t < eA; t += NT) { const int k = (int)(((float)t + 0.5f) * rA), i = t - k * nA; dst[t] = smL[i * 36 + k]; }
    for (int t = tid; t < 18 * nB; t += NT) { const int k = (int)(((float)t + 0.5f) * rB), i = t - k * nB; dst[eA + t] = smL[eA + i * 18 + k]; }
    for (int t = tid; t < 9 * nC; t += NT) { const int k = (int)(((float)t + 0.5f) * rC), i = t - k * nC; dst[eB + t] = smL[eB + i * 10 + k]; }
  }
  for (int e = tid; e < pm.ysize; e += NT) C.y[pm.y0 + e] = smY[e];
  SSLAM_FSTAMP(6)
  if (dbg && threadIdx.x == 0) { dbg[8] += 1; dbg[9] += nlv; dbg[10] += (sB[9] & 0xFFFF) + (sB[9] >> 16) + (h2 >> 20); dbg[11] += h2 & 0xFFFFF; }
}
#undef SSLAM_FSTAMP

template <int NT, bool RIGHT>
__global__ __launch_bounds__(NT, NT <= 256 ? 3 : 1) void k_front_pieces(BatchView V, CholView C, int begin, const int* __restrict__ idx) {
  extern __shared__ double sm[];
  const int q = idx ? idx[blockIdx.x] : begin + blockIdx.x;
  const PieceMeta pm = C.lpiece[q];
  front_piece<NT, RIGHT>(V, C, pm, C.lfgrp[q], sm, (C.dbg && blockIdx.x == 0) ? C.dbg + (pm.pad5 == 1 ? 48 : 16) : nullptr);
}

// top of the elimination tree: one workgroup per graph walks its remaining pieces in elimination order (like k_chol_tail)
template <int NT>
__global__ __launch_bounds__(NT) void k_front_tail(BatchView V, CholView C, const int* __restrict__ idx) {
  extern __shared__ double sm[];
  const int g = idx ? idx[blockIdx.x] : blockIdx.x;
  if (!V.lm[g].in_trial) return;
  const int q1 = C.tail_ptr[g + 1];
  for (int q = C.tail_ptr[g]; q < q1; ++q) {
    front_piece<NT, true>(V, C, C.lpiece[C.ltail0 + q], C.lfgrp[C.ltail0 + q], sm, (C.dbg && g == 0) ? C.dbg : nullptr);
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace sslam
