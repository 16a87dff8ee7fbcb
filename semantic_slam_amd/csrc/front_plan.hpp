// Front tables of the sparse block Cholesky: the piece plan of chol_plan.hpp once more, as ONE small blob of relative indices per
// workgroup instead of per-block / per-update records (round 6).
//
// Replaces, for large batches, the tables the factor kernels of the symbolic half of g2o's BlockSolverX + LinearSolverCSparse pair read
// (reference src/ps_graph_slam/graph_slam.cpp:67-73,199-205).
//
// Why.  The record plan spells every update  S(i,j) -= L(i,k) L(j,k)^T  out as an 8-byte record, every block of an update matrix as a
// 32-byte item plus an 8-byte record per child block, every block of L as a 32-byte record: 5.3 MB of tables per 5000-pose graph against
// 7.7 MB of L, read from HBM in three to four DEPENDENT trips per piece (piece record -> tables -> item -> child source -> child block),
// and the factorisation is bound by exactly those trips.  Everything those records say follows from the row
// structure of the piece's columns:
//   * a component (a connected subtree of the elimination tree, factored inside one workgroup) numbers its rows locally: its columns
//     0 .. nc-1 in elimination order, then its boundary rows (the structure of its root column) nc .. nc+nR-1;  at most 64 of them;
//   * the update matrix of a component is the DENSE lower triangle over its boundary rows -- it is a clique anyway -- stored block by
//     block, block row by block row: the offset of block (a, b) is arithmetic on two small per-row numbers;
//   * a child hands its update matrix to its parent through ONE byte per boundary row: the row's local number in the parent
//     ("relative index").  What the parent absorbs into a column and what it passes on through its own update matrix follows from that;
//   * which columns k update a target (i, j) is the AND of two 64-bit row masks (the columns that hold row i / row j), which the kernel
//     builds in LDS from the block list; where L(i,k) lives is a 16-bit entry of a (column, local row) map it builds as well.
// A workgroup's tables are one contiguous blob (a few hundred bytes to a few KB): [header | components | columns | blocks | levels |
// target tiles | boundary tables | child headers + their boundary tables], copied to LDS in one coalesced trip; 0.5 MB per 5000-pose
// graph.  The factor itself (Lval, y) keeps the layout of chol_plan.hpp, so the backward substitution, the marginals and the
// dependency-driven launches of small batches are untouched.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sslam {

struct FrontGrp { int blob, words, dbytes, graph; };   // word offset / length of the group's blob in fblob; bytes of the derived LDS tables; the graph (its LmState is fetched with the blob)

// blob header (12 words)
//   [0] ncomp | nlv << 8 | nchild << 16      [1] nc | nb << 16      [2] target tiles | (third and later child sources) << 20
//   [3] word offset of the columns   [4] of the blocks   [5] of the levels   [6] of the tiles (u16 each)   [7] of the child headers
//   [8] of the child sources   [9] first sources | second sources << 16   [10] word offset of the gather items (u16 each)   [11] gather items
constexpr int kFrontHdr = 12;
// child sources of the blocks (2 words each): [0] block | child (index among the component's children) << 16
//   [1] row index in the child's boundary | column index << 8 | rank << 16 (1: first source, ...); sorted by rank, then block
constexpr int kFrontMulti = 2;
// gather item (u16): block << 1 | half -- the rows 3 half .. 3 half + 2 of a block (a block of 3 rows has one item)
// component (8 words), from word 12 on
//   [0] nc | nR << 8 | T << 16 | nchild << 24 (T: 3-row tile rows of the boundary)   [1] Uval offset of its update matrix   [2] doubles in it (the
//   rhs part, 6 per boundary row, follows)   [3] word offset of its boundary table   [4] first child header (index) | tiles of the update matrices of the
//   components before it << 16   [5] [6] mask of the local rows of
//   dimension 6   [7] byte offset of its derived tables
constexpr int kFrontComp = 8;
// column (4 words):  [0] local offset of the diagonal block | local y offset << 16   [1] component | local column << 8 | (dim == 6) << 16 | level << 24
//   [2] offset in the unknown vector (rhs)   [3] first block (group-local; the diagonal one) | blocks << 14 | off-diagonal blocks with their row inside the component << 22
constexpr int kFrontCol = 4;
// block (4 words):   [0] H offset or -1 (fill)   [1] local L offset | local row << 16 | column (group-local) << 22 | transposed-in-H << 30 | diagonal << 31
//   [2] its first child source: child (index among the component's children) | row index in the child's boundary << 8 | column index << 14 | number of
//   child sources (capped at 255) << 20   [3] component | local column << 8 | (rows == 6) << 16 | (columns == 6) << 17
constexpr int kFrontBlk = 4;
// level (4 words):   [0] b0 | b1 << 16 (blocks of the level's columns, group-local)   [1] [2] target tiles [t0, t1)   [3] c0 | c1 << 16
constexpr int kFrontLv = 4;
// target tile (u16): block << 2 | tile row << 1 | tile column -- only tiles that receive an update; the upper right tile of a diagonal block is never listed
// boundary table entry (2 words):  [0] local row in the parent component | 6-rows before << 8 | 3-rows before << 16 | (dim == 6) << 24 | tag low 7 bits << 25
//   [1] offset of the block row (24 bits) | tag high 8 bits << 24     tag: the component (own tables) / the child header (copies) the entry belongs to
// child header (4 words):  [0] Uval offset   [1] doubles (rhs part follows)   [2] nR | component of the group << 8 | index among its children << 16
//   [3] word offset of the copy of its boundary table
constexpr int kFrontChild = 4;
constexpr int kFrontMaxRows = 64, kFrontMaxCols = 256, kFrontMaxComps = 255, kFrontMaxChildren = 255;

SSLAM_HD_INLINE int front_blk_doubles(int di, int dj) { return (di * dj + 1) & ~1; }
// offset of block (a, b), b <= a, inside an update matrix: rowbase[a] + what the blocks (a, 0 .. b-1) take
SSLAM_HD_INLINE int front_u_offset(int rowbase_a, int di, int n6_b, int n3_b) { return rowbase_a + n6_b * 6 * di + n3_b * (di == 6 ? 18 : 10); }
// derived tables of a component in LDS (bytes): [row masks 8 NR | child masks 8 NR (the children whose boundary holds local row r; the first 64 children) |
//   (column, row) -> L offset map 2 nc NR | y offsets 2 nc | tile rows T | child row maps nchild NR]
SSLAM_HD_INLINE int front_pad8(int x) { return (x + 7) & ~7; }
SSLAM_HD_INLINE int front_derived_bytes(int nc, int NR, int T, int nchild) {
  return 16 * NR + front_pad8(2 * nc * NR) + front_pad8(2 * nc) + front_pad8(T) + front_pad8(nchild * NR);
}

struct FrontIn {
  int ncol, npiece, ncomp;
  const std::vector<int>&bp, &brow, &boff, &bsrc;
  const std::vector<unsigned char>& bfmt;
  const std::vector<int>&col_comp, &col_piece, &col_dim, &col_xoff, &col_yoff, &col_il;
  const std::vector<int>& comp_parent;
  const std::vector<std::vector<int>>& comp_R;   // boundary rows (column ids, ascending)
};

struct FrontHost {
  bool ok = false;
  std::string why;                 // why the plan has no front tables (a component with more than 64 local rows, ...)
  std::vector<uint32_t> blob;
  std::vector<FrontGrp> grp;       // by piece id
  std::vector<int> lds;            // LDS doubles of the factor kernel, by piece id
  int64_t unz = 0;                 // doubles of update matrices in the front layout
};

// pieces: PieceMeta records (c0, nc, b0, nb, lbase, lsize, y0, ysize); ilv: the levels of every piece (global column / block ids)
template <class PieceVec, class LevelVec>
inline void front_build(const FrontIn& in, const PieceVec& piece, const LevelVec& ilv, FrontHost& out) {
  out = FrontHost();
  const int ncomp = in.ncomp;
  // components: columns (ascending), children (ascending), local numbering
  std::vector<int> c_ptr(ncomp + 1, 0), c_cols(in.ncol), col_kc(in.ncol, 0);
  for (int j = 0; j < in.ncol; ++j) c_ptr[in.col_comp[j] + 1]++;
  for (int c = 0; c < ncomp; ++c) c_ptr[c + 1] += c_ptr[c];
  {
    std::vector<int> cur(c_ptr.begin(), c_ptr.end() - 1);
    for (int j = 0; j < in.ncol; ++j) { const int c = in.col_comp[j]; col_kc[j] = cur[c] - c_ptr[c]; c_cols[cur[c]++] = j; }
  }
  std::vector<int> k_ptr(ncomp + 1, 0), k_idx(ncomp);
  for (int c = 0; c < ncomp; ++c) if (in.comp_parent[c] >= 0) k_ptr[in.comp_parent[c] + 1]++;
  for (int c = 0; c < ncomp; ++c) k_ptr[c + 1] += k_ptr[c];
  {
    std::vector<int> cur(k_ptr.begin(), k_ptr.end() - 1);
    for (int c = 0; c < ncomp; ++c) if (in.comp_parent[c] >= 0) k_idx[cur[in.comp_parent[c]]++] = c;
  }
  auto fail = [&](const char* w) { out.ok = false; out.why = w; out.blob.clear(); out.grp.clear(); out.lds.clear(); };
  // boundary tables + update-matrix storage of every component
  struct BT { uint32_t w0; uint32_t rowbase; };
  std::vector<int> bt_ptr(ncomp + 1, 0), c_ubase(ncomp, 0), c_usize(ncomp, 0), c_T(ncomp, 0);
  for (int c = 0; c < ncomp; ++c) bt_ptr[c + 1] = bt_ptr[c] + (int)in.comp_R[c].size();
  std::vector<BT> bt(bt_ptr[ncomp]);
  int64_t ucur = 0;
  for (int c = 0; c < ncomp; ++c) {
    const std::vector<int>& R = in.comp_R[c];
    const int nc = c_ptr[c + 1] - c_ptr[c], nR = (int)R.size();
    if (nc + nR > kFrontMaxRows) return fail("a component has more than 64 local rows");
    if (nc > 255 || k_ptr[c + 1] - k_ptr[c] > kFrontMaxChildren) return fail("a component has too many columns or children");
    const int par = in.comp_parent[c];
    if (nR > 0 && par < 0) return fail("a root component has boundary rows");
    int n6 = 0, n3 = 0, rowbase = 0;
    for (int a = 0; a < nR; ++a) {
      const int g = R[a], d = in.col_dim[g];
      int rel;
      if (in.col_comp[g] == par) rel = col_kc[g];
      else {
        const std::vector<int>& Rp = in.comp_R[par];
        const auto it = std::lower_bound(Rp.begin(), Rp.end(), g);
        if (it == Rp.end() || *it != g) return fail("a boundary row is missing from the parent's boundary");
        rel = (c_ptr[par + 1] - c_ptr[par]) + (int)(it - Rp.begin());
      }
      bt[bt_ptr[c] + a] = BT{(uint32_t)rel | ((uint32_t)n6 << 8) | ((uint32_t)n3 << 16) | ((uint32_t)(d == 6) << 24), (uint32_t)rowbase};
      rowbase += n6 * 6 * d + n3 * (d == 6 ? 18 : 10) + front_blk_doubles(d, d);   // blocks (a, 0 .. a)
      if (d == 6) ++n6; else ++n3;
    }
    if (rowbase >= (1 << 24)) return fail("an update matrix is too large for the packed boundary tables");
    c_T[c] = 2 * n6 + n3;
    c_usize[c] = rowbase;
    c_ubase[c] = (int)ucur;
    ucur += rowbase + 6 * nR;
    if (ucur >= ((int64_t)1 << 31) - 4096) return fail("update matrices too large for int32 offsets");
  }
  out.unz = ucur;
  // one blob per group
  out.grp.assign(in.npiece, FrontGrp{0, 0, 0, 0});
  out.lds.assign(in.npiece, 0);
  std::vector<int> comps, lcomp(ncomp, -1);
  std::vector<uint64_t> rw;          // row masks of the group's components, [component of the group][64]
  struct MultiRec { int rank, blk, f, qa, qb; };
  std::vector<MultiRec> multi;
  for (int p = 0; p < in.npiece; ++p) {
    const auto& pm = piece[p];
    if (pm.nc > kFrontMaxCols || pm.nb >= (1 << 13) || pm.lsize >= (1 << 16) || pm.ysize >= (1 << 16)) return fail("a piece is too large for the packed front tables");
    comps.clear();
    for (int j = pm.c0; j < pm.c0 + pm.nc; ++j) { const int c = in.col_comp[j]; if (lcomp[c] < 0) { lcomp[c] = 0; comps.push_back(c); } }
    std::sort(comps.begin(), comps.end());
    if ((int)comps.size() > kFrontMaxComps) return fail("a group has more than 255 components");
    for (size_t q = 0; q < comps.size(); ++q) lcomp[comps[q]] = (int)q;
    const int ncg = (int)comps.size();
    int nchild = 0;
    for (int c : comps) nchild += k_ptr[c + 1] - k_ptr[c];
    if (nchild > 32767) return fail("a group has too many child components");
    // row masks
    rw.assign((size_t)ncg * 64, 0);
    auto lrow = [&](int c, int g) -> int {   // local number of row g (a column id) in component c
      if (in.col_comp[g] == c) return col_kc[g];
      const std::vector<int>& R = in.comp_R[c];
      return (c_ptr[c + 1] - c_ptr[c]) + (int)(std::lower_bound(R.begin(), R.end(), g) - R.begin());
    };
    for (int j = pm.c0; j < pm.c0 + pm.nc; ++j) {
      const int c = in.col_comp[j], q = lcomp[c];
      for (int t = in.bp[j] + 1; t < in.bp[j + 1]; ++t) rw[(size_t)q * 64 + lrow(c, in.brow[t])] |= 1ull << col_kc[j];
    }
    const int w0 = (int)out.blob.size();
    auto& B = out.blob;
    B.resize(w0 + kFrontHdr + kFrontComp * ncg, 0);
    // columns
    const int w_cols = (int)B.size() - w0;
    for (int j = pm.c0; j < pm.c0 + pm.nc; ++j) {
      const int c = in.col_comp[j];
      const int doff = in.boff[in.bp[j]] - pm.lbase, yl = in.col_yoff[j] - pm.y0;
      if (in.col_il[j] > 255) return fail("a piece has more than 255 internal levels");
      B.push_back((uint32_t)doff | ((uint32_t)yl << 16));
      B.push_back((uint32_t)lcomp[c] | ((uint32_t)col_kc[j] << 8) | ((uint32_t)(in.col_dim[j] == 6) << 16) | ((uint32_t)in.col_il[j] << 24));
      B.push_back((uint32_t)in.col_xoff[j]);
      int mi = 0;
      for (int t = in.bp[j] + 1; t < in.bp[j + 1]; ++t) if (in.col_comp[in.brow[t]] == c) ++mi;
      B.push_back((uint32_t)(in.bp[j] - pm.b0) | ((uint32_t)(in.bp[j + 1] - in.bp[j]) << 14) | ((uint32_t)mi << 22));
    }
    // blocks
    multi.clear();
    const int w_blk = (int)B.size() - w0;
    for (int j = pm.c0; j < pm.c0 + pm.nc; ++j) {
      const int c = in.col_comp[j];
      for (int t = in.bp[j]; t < in.bp[j + 1]; ++t) {
        const int lr = lrow(c, in.brow[t]);
        B.push_back((uint32_t)in.bsrc[t]);
        B.push_back((uint32_t)(in.boff[t] - pm.lbase) | ((uint32_t)lr << 16) | ((uint32_t)(j - pm.c0) << 22) | ((uint32_t)(in.bfmt[t] ? 1 : 0) << 30) | ((uint32_t)(t == in.bp[j]) << 31));
        // child sources: the children whose boundary holds both the row and the column (the first in the block record, the others listed)
        int nsrc = 0, f = 0, qa = 0, qb = 0;
        for (int kq = k_ptr[c]; kq < k_ptr[c + 1]; ++kq) {
          const std::vector<int>& Rd = in.comp_R[k_idx[kq]];
          const auto ia = std::lower_bound(Rd.begin(), Rd.end(), in.brow[t]);
          if (ia == Rd.end() || *ia != in.brow[t]) continue;
          const auto ib = std::lower_bound(Rd.begin(), Rd.end(), j);
          if (ib == Rd.end() || *ib != j) continue;
          if (nsrc++ == 0) { f = kq - k_ptr[c]; qa = (int)(ia - Rd.begin()); qb = (int)(ib - Rd.begin()); }
          multi.push_back(MultiRec{nsrc, t - pm.b0, kq - k_ptr[c], (int)(ia - Rd.begin()), (int)(ib - Rd.begin())});
        }
        B.push_back((uint32_t)f | ((uint32_t)qa << 8) | ((uint32_t)qb << 14) | ((uint32_t)std::min(nsrc, 255) << 20));
        B.push_back((uint32_t)lcomp[c] | ((uint32_t)col_kc[j] << 8) | ((uint32_t)(in.col_dim[in.brow[t]] == 6) << 16) | ((uint32_t)(in.col_dim[j] == 6) << 17));
      }
    }
    // levels + target tiles
    const int w_lv = (int)B.size() - w0;
    std::vector<uint16_t> tiles;
    for (int l = 0; l < pm.nilv; ++l) {
      const auto& lv = ilv[pm.ilv0 + l];
      const int t0 = (int)tiles.size();
      for (int j = lv.c0; j < lv.c1; ++j) {
        const int c = in.col_comp[j], q = lcomp[c], lj = col_kc[j], dj = in.col_dim[j];
        for (int t = in.bp[j]; t < in.bp[j + 1]; ++t) {
          const int li = lrow(c, in.brow[t]);
          const bool diag = t == in.bp[j];
          const uint64_t m = diag ? rw[(size_t)q * 64 + lj] : (rw[(size_t)q * 64 + li] & rw[(size_t)q * 64 + lj]);
          if (!m) continue;
          const int di = in.col_dim[in.brow[t]];
          for (int tr = 0; tr < di / 3; ++tr)
            for (int tc = 0; tc < dj / 3; ++tc) {
              if (diag && tc > tr) continue;
              tiles.push_back((uint16_t)(((t - pm.b0) << 2) | (tr << 1) | tc));
            }
        }
      }
      B.push_back((uint32_t)(lv.b0 - pm.b0) | ((uint32_t)(lv.b1 - pm.b0) << 16));
      B.push_back((uint32_t)t0);
      B.push_back((uint32_t)tiles.size());
      B.push_back((uint32_t)(lv.c0 - pm.c0) | ((uint32_t)(lv.c1 - pm.c0) << 16));
    }
    const int w_tile = (int)B.size() - w0;
    for (size_t q = 0; q < tiles.size(); q += 2) B.push_back((uint32_t)tiles[q] | ((uint32_t)(q + 1 < tiles.size() ? tiles[q + 1] : 0) << 16));
    // own boundary tables, then child headers, then the children's boundary tables
    std::vector<int> w_bt(ncg, 0);
    for (int q = 0; q < ncg; ++q) {
      const int c = comps[q];
      w_bt[q] = (int)B.size() - w0;
      for (int a = bt_ptr[c]; a < bt_ptr[c + 1]; ++a) { B.push_back(bt[a].w0 | ((uint32_t)(q & 127) << 25)); B.push_back(bt[a].rowbase | ((uint32_t)(q >> 7) << 24)); }
    }
    const int w_child = (int)B.size() - w0;
    const size_t hdr0 = B.size();
    B.resize(B.size() + (size_t)kFrontChild * nchild, 0);
    {
      int ch = 0;
      for (int q = 0; q < ncg; ++q) {
        const int c = comps[q];
        for (int kq = k_ptr[c]; kq < k_ptr[c + 1]; ++kq, ++ch) {
          const int d = k_idx[kq];
          const int nRd = bt_ptr[d + 1] - bt_ptr[d];
          const uint32_t wb = (uint32_t)(B.size() - w0);
          B[hdr0 + (size_t)kFrontChild * ch + 0] = (uint32_t)c_ubase[d];
          B[hdr0 + (size_t)kFrontChild * ch + 1] = (uint32_t)c_usize[d];
          B[hdr0 + (size_t)kFrontChild * ch + 2] = (uint32_t)nRd | ((uint32_t)q << 8) | ((uint32_t)(kq - k_ptr[c]) << 16);
          B[hdr0 + (size_t)kFrontChild * ch + 3] = wb;
          for (int a = bt_ptr[d]; a < bt_ptr[d + 1]; ++a) { B.push_back(bt[a].w0 | ((uint32_t)(ch & 127) << 25)); B.push_back(bt[a].rowbase | ((uint32_t)(ch >> 7) << 24)); }
        }
      }
    }
    // further child sources, by rank
    const int w_multi = (int)B.size() - w0;
    std::stable_sort(multi.begin(), multi.end(), [](const MultiRec& a, const MultiRec& b2) { return a.rank < b2.rank; });
    int n1 = 0, n2 = 0;
    for (const MultiRec& r : multi) {
      if (r.rank == 1) ++n1;
      if (r.rank == 2) ++n2;
      B.push_back((uint32_t)r.blk | ((uint32_t)r.f << 16));
      B.push_back((uint32_t)r.qa | ((uint32_t)r.qb << 8) | ((uint32_t)std::min(r.rank, 65535) << 16));
    }
    if (multi.size() > 65535) return fail("a group has too many child sources");
    // gather items: the 3-row halves of every block
    const int w_items = (int)B.size() - w0;
    int nitems = 0;
    {
      std::vector<uint16_t> items;
      for (int j = pm.c0; j < pm.c0 + pm.nc; ++j)
        for (int t = in.bp[j]; t < in.bp[j + 1]; ++t)
          for (int h = 0; h < in.col_dim[in.brow[t]] / 3; ++h) items.push_back((uint16_t)(((t - pm.b0) << 1) | h));
      nitems = (int)items.size();
      for (size_t q = 0; q < items.size(); q += 2) B.push_back((uint32_t)items[q] | ((uint32_t)(q + 1 < items.size() ? items[q + 1] : 0) << 16));
    }
    while ((B.size() - w0) & 3) B.push_back(0);   // whole 16-byte pieces
    // components
    int dbytes = 0, child0 = 0, tcum = 0;
    for (int q = 0; q < ncg; ++q) {
      const int c = comps[q];
      const int nc = c_ptr[c + 1] - c_ptr[c], nR = bt_ptr[c + 1] - bt_ptr[c], nch = k_ptr[c + 1] - k_ptr[c];
      uint64_t p6 = 0;
      for (int k = 0; k < nc; ++k) if (in.col_dim[c_cols[c_ptr[c] + k]] == 6) p6 |= 1ull << k;
      for (int a = 0; a < nR; ++a) if (in.col_dim[in.comp_R[c][a]] == 6) p6 |= 1ull << (nc + a);
      uint32_t* W = B.data() + w0 + kFrontHdr + kFrontComp * q;
      W[0] = (uint32_t)nc | ((uint32_t)nR << 8) | ((uint32_t)c_T[c] << 16) | ((uint32_t)nch << 24);
      W[1] = (uint32_t)c_ubase[c]; W[2] = (uint32_t)c_usize[c]; W[3] = (uint32_t)w_bt[q]; W[4] = (uint32_t)child0 | ((uint32_t)tcum << 16);
      W[5] = (uint32_t)(p6 & 0xFFFFFFFFu); W[6] = (uint32_t)(p6 >> 32); W[7] = (uint32_t)dbytes;
      dbytes += front_derived_bytes(nc, nc + nR, c_T[c], nch);
      child0 += nch;
      tcum += c_T[c] * (c_T[c] + 1) / 2;
      if (tcum > 65535) return fail("a group has too many update-matrix tiles");
    }
    uint32_t* Hd = B.data() + w0;
    Hd[0] = (uint32_t)ncg | ((uint32_t)pm.nilv << 8) | ((uint32_t)nchild << 16);
    Hd[1] = (uint32_t)pm.nc | ((uint32_t)pm.nb << 16);
    if ((int)multi.size() - n1 - n2 > 4095 || tiles.size() >= (1u << 20)) return fail("a group has too many child sources or target tiles");
    Hd[2] = (uint32_t)tiles.size() | ((uint32_t)(multi.size() - n1 - n2) << 20);
    Hd[8] = (uint32_t)w_multi; Hd[9] = (uint32_t)n1 | ((uint32_t)n2 << 16);
    Hd[10] = (uint32_t)w_items; Hd[11] = (uint32_t)nitems;
    Hd[3] = (uint32_t)w_cols; Hd[4] = (uint32_t)w_blk; Hd[5] = (uint32_t)w_lv; Hd[6] = (uint32_t)w_tile; Hd[7] = (uint32_t)w_child;
    const int words = (int)B.size() - w0;
    const int lds = ((pm.lsize + 1) & ~1) + ((pm.ysize + 1) & ~1) + words / 2 + (dbytes + 7) / 8 + 2;
    out.grp[p] = FrontGrp{w0, words, dbytes, pm.graph};
    out.lds[p] = lds;
    for (int c : comps) lcomp[c] = -1;
  }
  out.ok = true;
}

}  // namespace sslam
