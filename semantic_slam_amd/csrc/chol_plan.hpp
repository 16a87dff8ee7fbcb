// Host-side symbolic phase of the sparse block Cholesky (no HIP types: also compiled into the CPU-only plan tests).
//
// Replaces the symbolic half of g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with "lm_var"
// (reference src/ps_graph_slam/graph_slam.cpp:27,67-73).
//
// MI355X design.  The block elimination tree of a SLAM graph is bushy at the bottom and a long chain at the top
// (151 levels for the 5000-pose / 1000-landmark graph), and one numeric factorisation is only ~25 MFLOP / 7.5 MB per
// graph: a launch per level is all latency, and so is every dependent trip to HBM.  The tree is therefore cut into
// *pieces* -- connected sets of columns (a subtree minus the pieces hanging below it) whose part of L fits in the LDS of
// one workgroup -- and the pieces talk to each other the multifrontal way:
//   * a piece is factored by ONE workgroup.  Its columns are gathered into LDS (A + lambda I minus the update-matrix
//     blocks its child pieces left for it: coalesced 288-byte reads, no arithmetic), the levels *inside* the piece run
//     out of LDS with workgroup barriers between them, and L leaves as one contiguous stream;
//   * every update  S(i,j) -= L(i,k) L(j,k)^T  is computed exactly once, in the piece that owns the SOURCE column k, from
//     LDS.  Updates whose target column lies in a higher piece are summed into the piece's *update matrix* U over its
//     boundary rows (plus whatever its children handed up for the same block) and written to HBM once; the parent either
//     absorbs a U block into its own column or passes it further up;
//   * pieces of equal depth in the piece tree share a launch (a handful of launches instead of 151); once a graph is
//     down to a few pieces per depth, the rest ("tail") is walked by one workgroup per graph in a single launch;
//   * L is laid out piece by piece, columns of a piece by internal level;
//   * updates are cut into work items of <= chunk consecutive updates of ONE target block; an item is executed by
//     four lanes holding the 3x3 tiles of the target.  A target with a single item is finished in place ("sole"),
//     longer lists go through per-item partial tiles that are summed in item order -> deterministic.
// The backward substitution walks the same pieces top-down.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <queue>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace sslam {
#ifdef SSLAM_PLAN_TIMERS
#include <chrono>
#define SSLAM_PT_INIT auto pt_ = std::chrono::steady_clock::now();
#define SSLAM_PT(name) { auto n_ = std::chrono::steady_clock::now(); fprintf(stderr, "[plan-timer] %-14s %.3f ms\n", name, std::chrono::duration<double, std::milli>(n_ - pt_).count()); pt_ = n_; }
#else
#define SSLAM_PT_INIT
#define SSLAM_PT(name)
#endif

struct ColMeta { int xoff, yoff, dim, graph, b0, nb, nbi, base, f0, f1, piece, ilevel; };
// xoff: offset in the unknown vector (internal row order); yoff: offset in elimination order (the forward-substituted
// rhs y lives in that order so that a piece's y is contiguous); blocks [b0, b0 + nb), diagonal first, the first nbi
// (diagonal included) have their row inside the column's own piece; base = Lval offset of the diagonal block;
// [f0, f1) = the blocks of ROW j (FwdMeta), for the forward substitution of the multi right-hand-side solves

struct BlkMeta { int off, src, xoff_row, yoff_row, coldiag, colyoff, info, as0; };
// off: Lval offset; src: H offset or -1; x / y offsets of the block's row; Lval offset of the diagonal block and y offset of
// the block's column; info = di | dj << 4 | fmt << 8 | diag << 9 | row-in-piece << 10 | nas << 16; [as0, as0 + nas) = the child
// update-matrix blocks to subtract when the block is gathered (piece-local index into the piece's AsmSrc records)
constexpr int kBlkFmt = 1 << 8, kBlkDiag = 1 << 9, kBlkRowIn = 1 << 10, kBlkNasShift = 16;
// rank of a row-in-piece block among the piece's row-in-piece blocks (the slot the backward substitution keeps it in): low 5 bits at 11,
// high 8 bits at 24
inline int blk_irank_bits(int r) { return ((r & 31) << 11) | ((r >> 5) << 24); }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int blk_irank(int info) { return ((info >> 11) & 31) | (((info >> 24) & 255) << 5); }

struct AsmSrc { int uoff, uyoff; };      // an update-matrix block of a child piece (Uval offset); uyoff >= 0: its rhs part too (diagonal blocks)
struct FwdMeta { int off, yoff; };       // block L(j,k) of row j: Lval offset | (dim k == 6) << 31, y offset of column k

// One update  S(i,j) -= L(i,k) L(j,k)^T  in 8 bytes (round 5; rounds 1-4: four ints, and the update records were a third of a plan's table
// bytes and as large in a piece's LDS as its part of L).  Both operands are blocks of the piece that owns the source column k, so their
// offsets are piece-local and fit 16 bits:
//   ab = local Lval offset of L_ik | local offset of L_jk << 16
//   xk = local y offset of column k (13 bits) | flags << 28                                       target-major lists (upd[])
//      = local offset of the target block (16 bits) | local y offset of its column << 16 (12 bits) | flags << 28     right-looking lists (rupd[])
struct UpdMeta { unsigned ab, xk; };
constexpr unsigned kUpdDi6 = 1u << 28;    // target block has 6 rows (else 3)
constexpr unsigned kUpdDk6 = 1u << 29;    // source column k is 6 wide (else 3)
constexpr unsigned kUpdDiag = 1u << 30;   // target is a diagonal block (carries the forward-substitution rhs too)
constexpr unsigned kUpdDj6 = 1u << 31;    // target column is 6 wide (else 3)
constexpr int kUpdLocalMax = 1 << 16, kUpdYMax = 1 << 13, kUpdRightYMax = 1 << 12;
#if defined(__HIPCC__)
#define SSLAM_HD_INLINE __host__ __device__ inline
#else
#define SSLAM_HD_INLINE inline
#endif
SSLAM_HD_INLINE int upd_ua(const UpdMeta& r) { return (int)(r.ab & 0xFFFFu); }
SSLAM_HD_INLINE int upd_ub(const UpdMeta& r) { return (int)(r.ab >> 16); }
SSLAM_HD_INLINE int upd_yk(const UpdMeta& r) { return (int)(r.xk & 0x1FFFu); }            // target-major records
SSLAM_HD_INLINE int upd_rt(const UpdMeta& r) { return (int)(r.xk & 0xFFFFu); }            // right-looking records: target block
SSLAM_HD_INLINE int upd_ry(const UpdMeta& r) { return (int)((r.xk >> 16) & 0xFFFu); }     // ... and the local y offset of its column
inline UpdMeta upd_make(int ua_local, int ub_local, int yk_local, unsigned flags) { return UpdMeta{(unsigned)ua_local | ((unsigned)ub_local << 16), (unsigned)yk_local | flags}; }
inline UpdMeta upd_make_right(int ua_local, int ub_local, int tl, int yl, unsigned flags) { return UpdMeta{(unsigned)ua_local | ((unsigned)ub_local << 16), (unsigned)tl | ((unsigned)yl << 16) | flags}; }

}  // namespace sslam
#include "front_plan.hpp"
namespace sslam {

struct ItemMeta { int u0, n, tloff, flags; };   // updates [u0, u0 + n) (index into the piece's update records, LDS copy) of one target
                                                // block at piece-local offset tloff
constexpr int kItemSole = 1;                     // flags: bit 0 sole (subtract in place); bits 1..11 partial slot; bits 12.. local y offset
constexpr int kItemSlotShift = 1, kItemSlotMask = 0x7FF, kItemYShift = 12;
struct MbMeta { int tloff, ps0, n, info; };      // a target block with n > 1 items: partial slots [ps0, ps0 + n); info = di | dj << 4 | diag << 9 | ylocal << 12
struct RCol { int u0, n; };             // tail pieces: the internal updates a finished column applies to later columns of its piece (right-looking form),
                                         // records [u0, u0 + n) of rupd[] counted from the piece's first record (PieceMeta.pad3), in UpdMeta's
                                         // right-looking form
struct ILevel { int c0, c1, b0, b1, it0, it1, mb0, mb1; };   // one level inside a piece: columns, blocks (global ids), items and multi-blocks (piece-local)

// update-matrix side: a block U(a,b) of the piece = sum of its own updates [u0, u0 + n) (upd[], sources in the piece) + the
// child blocks [s0, s0 + ns) (usrc[]); written to Uval[uoff ...] (and the rhs part to Uval[uyoff ...] for a diagonal block)
struct UItem { int u0, n, uoff, flags, s0, ns, uyoff, pad; };   // flags: bit 0 sole; bits 1..11 partial slot; bit 12 di == 6; bit 13 dj == 6; bit 14 diag
constexpr int kUItemDi6 = 1 << 12, kUItemDj6 = 1 << 13, kUItemDiag = 1 << 14;
struct UMb { int uoff, ps0, n, info, s0, ns, uyoff, pad; };     // a U block whose own list was split: info = di | dj << 4 | diag << 9

struct PieceMeta { int graph, c0, nc, b0, nb, lbase, lsize, y0, ysize, ilv0, nilv, iit0, nit_i, iu0, nu_i, imb0, nimb, as0, nas, uit0, nuit, umb0, numb,
                   uu0, nuu, us0, nus, n36, n18, nint, pad3, pad4, nu4, nu2, nu1, pad5; };
// nu4 / nu2 / nu1 (round 4): the piece's U items are ordered [sole items of 6 x 6 blocks | sole 6 x 3 and 3 x 6 | sole 3 x 3 | items of split
// lists]; a sole item of nu4 is four 3 x 3 tiles, of nu2 two, of nu1 one -- the update-matrix phase hands the TILES to the lanes (a quad per
// item left three of four lanes idle on a 3 x 3 block, and 3 x 3 landmark-landmark blocks are most of a leaf piece's update matrix)
// inside the piece (all copied to LDS when the piece starts, so that its levels never wait for HBM): levels [ilv0, +nilv), items
// [iit0, +nit_i), update records [iu0, +nu_i), multi-blocks [imb0, +nimb), assembly sources [as0, +nas);
// update matrix: U items [uit0, +nuit), split U blocks [umb0, +numb), their update records [uu0, +nuu) and child sources [us0, +nus)
// (UItem.u0 / .s0 and UMb.s0 are relative to uu0 / us0: the per-depth kernels stage these records in LDS as well)

// Factor storage of a piece.  Flat form (LDS, and HBM when CholView::flat_L is set for the multi right-hand-side kernels): the
// piece's blocks sorted by size class, [n36 blocks of 36 doubles | n18 of 18 | the rest of 10], each block row-major.  HBM form of
// the LM loop: every class transposed -- element k of the i-th block of a class at  class start + k * (blocks in the class) + i --
// so that the backward substitution, which runs one thread per block and touches every element once, reads with consecutive lanes
// on consecutive addresses and needs no LDS staging of the factor.
// storage of a di x dj block: 3 x 3 blocks are padded to 10 doubles so that every block (and every row of a 6-wide block) starts
// on a 16-byte boundary (ds_read_b128 / global_load_dwordx4 in the update kernels)
inline int blk_doubles(int di, int dj) { return (di * dj + 1) & ~1; }
constexpr int kItemDoubles = 42;     // LDS doubles per partial tile: 6 x 6 entries + 6 rhs components
constexpr int kMaxILevels = 64;      // internal levels per piece (LDS table in the kernels)

struct SymGraph { int prow0, nprow, lrow0, nlrow; int pp0 = 0, pp1 = 0, pl0 = 0, pl1 = 0, ll0 = 0, ll1 = 0; };   // pp / pl / ll: the graph's range of ppoff / plblk / llblk
struct SymIn {
  int B = 0, nPr = 0, nLr = 0;
  std::vector<SymGraph> seg;
  std::vector<std::pair<int, int>> ppoff, plblk;   // unique off-diagonal blocks: (pose row a < pose row b), (pose row, landmark row)
  std::vector<std::pair<int, int>> llblk;          // (landmark row a < landmark row b): point-point edges
  int64_t hll_base = 0, hpp_off_base = 0, hpl_base = 0, hll_off_base = 0;
};
struct CholOpts {
  int cap_leaf = -1;       // doubles of L per piece (pieces that share launches): small pieces, many resident per CU.  -1: 700 for batches >= 32, else 900
  int cap_tail = 4608;     // doubles of L per piece of a tail (two tail workgroups per CU must fit the LDS)
  int max_blocks = 224;    // blocks per piece
  int tail_width = -1;     // a graph's tail starts where it has <= tail_width pieces per depth; -1: 2; 0: no tail.  (Rounds 3-5 took 6 for batches >= 32;
                           // since the mid pieces run on 128 threads the depths below a short tail are cheaper in the mid class: 1024 L graphs, factor /
                           // backward solve 12.32 / 4.40 ms at 6, 12.15 / 4.33 at 4, 12.14 / 4.29 at 3, 12.05 / 4.25 at 2, 12.13 / 4.27 at 1, 12.38 / 4.36 without a tail)
  // Mid class (round 5).  Above the bushy bottom a column has 15-25 blocks and a leaf-sized piece holds two of them: its update matrix
  // (boundary^2: 160-190 blocks) is 4-5 x its part of L, every such piece reads its children's and writes its own, and ONE wave walks
  // those ~600 tiles in ten passes of dependent HBM trips (~100 us per piece against ~20 us for a leaf piece; 53 % of all update-matrix
  // bytes).  The depths between the bottom and the tail -- where a graph has <= mid_width pieces per depth -- are cut again with a larger
  // cap and run by wider workgroups, one launch per depth next to the leaf pieces of that depth: several columns of a chain share a
  // piece, their updates stay in LDS (right-looking lists like the tail's), and four waves share the tiles of what is still handed up.
  int mid_width = -1;      // -1: 60 for batches >= 32, else 0 (small batches run k_chol_flow with one workgroup size); 0: no mid class
  int cap_mid = 1800;      // doubles of L per mid piece (512 L graphs: 1800 / 2400 / 3600 -> 9.14 / 9.47 / 10.2 ms per factorisation, 9.55 without the class)
  int nt_mid = 128;        // its workgroup.  Round 6, front kernels, 512 L graphs, mid launches per factorisation: 512 threads 3.87 ms, 256 1.91, 128 1.54, 64 1.89
                           // (a piece's phases hold a few hundred lanes of work at most; every further wave adds its share of the barriers and idle passes)
  int nt_ftail = 256;      // workgroup of k_front_tail (the front tables do not depend on it): 128 | 256 | 512 | 1024; tail launch 1.12 / 0.84 / 1.13 ms at 128 / 256 / 512
  int pcap_mid = 16;
  int nt_bleaf = 0, nt_bmid = 0, nt_btail = 0;   // workgroups of the backward-substitution launches (0: those of the factorisation; the kernels do not depend on the cut)
  int nt_leaf = -1, nt_tail = 512;    // workgroup sizes the items are cut for; nt_leaf -1: 128 for batches >= 32 (groups of pieces, below), else 64
  int min_chunk = 4;       // a list of <= min_chunk updates is never split
  int split_min = 4096;    // a depth with at least this many pieces is launched in up to four parts, by LDS need
  int pcap_leaf = 4, pcap_tail = 32;   // partial tiles per phase (split lists): LDS budget of a piece
  // Groups (round 5 default for large batches): the leaf pieces of equal depth of a graph are packed, neighbours in elimination order, into
  // groups of <= group_cap doubles of L (and <= group_blocks blocks) that ONE workgroup factors side by side -- the levels inside the
  // members run together, so every dependent step (a round trip to HBM, a 6 x 6 factor by one lane, a barrier) serves several pieces, and a
  // wave's lanes are filled (a lone leaf piece keeps ~40 % of 64 lanes busy in its parallel phases and one or two in its diagonal blocks).
  // 512 L graphs, factor / backward solve per batch: no groups 8.87 / 3.35 ms; 64-thread groups of 1000 (cap_leaf 500) 8.24 / 2.69;
  // 128-thread groups of 2800 (cap_leaf 700) 7.9-8.3 / 2.3 (the default); 256-thread groups of 4000-5000 8.5-9.3 / 2.3-2.4.
  int group_cap = -1;      // -1: 2800 for batches >= 32, else 0 (no groups: small batches run the dependency-driven launch, one piece per workgroup)
  int group_blocks = 1024;
  int ustage = -1;         // 1: the per-depth kernels stage the update-matrix records in LDS too (one round trip for all tables: shorter
                           //    piece latency, fewer pieces per CU); -1: 1 for batches < 32 (latency-bound), else 0 (residency-bound)
  int order = 1;           // 0: minimum degree, lowest index first (rounds 1-3: eliminates a pose chain from one end -> a tree as deep as the chain);
                           // 1: multiple minimum degree over independent sets (below): a chain halves per round -> O(log n) levels.  Small
                           // batches are latency-bound on the depth of the tree (one 5000-pose graph 588 -> 1050 LM iterations/s); on the
                           // 512-graph batch the two orders took the same time while a quad per update-matrix item left most lanes idle on
                           // the wider tree's many 3 x 3 blocks, and since the items run by tiles the shallow tree is the faster one there
                           // too (9.30 vs 9.77 ms per 512 factorisations) -- at 26 % more update-matrix bytes through HBM
  double order_mul = -1;   // a round eliminates an independent set of the nodes with degree <= order_mul * (minimum degree) + order_add;
  int order_add = -1;      // -1: (2.0, 4) for batches < 32 (shallowest tree: latency), (1.5, 2) for larger ones (less fill and smaller update
                           // matrices, five levels more: throughput -- measured 9.30 vs 9.48 ms per 512 factorisations)
  int order_bits_max = 2048;   // graphs up to this many nodes are ordered on adjacency bitsets (same order, a fraction of the time); 0: never
  bool dump = false;
  int front = -1;          // front tables (front_plan.hpp: one blob of relative indices per workgroup; front_kernels.hpp): 1: built, and every
                           // factorisation -- per-depth launches and the dependency-driven launches alike -- runs through them; 0: the record
                           // kernels (chol_piece); -1: 1 for batches >= 32, else 0.  Measured (round 6): 512 L graphs 7.85 vs 8.0 ms per
                           // factorisation with a fifth of the table bytes; the orchestrator's tick (latency-bound: the LDS digest of the blob
                           // sits on the chain of pieces) 5.53-5.70 vs 5.27-5.30 ms at 110 keyframes, 8.0 vs 7.64 at 436 -> record kernels there
  int flow = 1;            // small batches: 0 a launch per depth; 1 the dependency-driven single launch (k_chol_flow) when the tree is narrower than its
                           // grid; 2 also on wide trees (per-depth launches for the bottom, measured slower: tests only)
  // SSLAM_CHOL_OPTS="key=value,key=value,...": every plan option above by its field name (tests force the piece shapes of a 5000-pose graph
  // onto small graphs with it; tuning sweeps), plus order=mmd|mindeg and dump=1.  The ONE environment switch of the plan.
  void from_env() {
    {   // legacy per-knob variables (rounds 2-4) are ignored: say so once
      static bool warned = false;
      if (!warned) {
        warned = true;
        static const char* legacy[] = {"SSLAM_CHOL_CAP_LEAF", "SSLAM_CHOL_CAP_TAIL", "SSLAM_CHOL_TAIL_WIDTH", "SSLAM_CHOL_NT_TAIL", "SSLAM_CHOL_NT_LEAF", "SSLAM_CHOL_MAX_BLOCKS",
                                       "SSLAM_CHOL_DUMP", "SSLAM_CHOL_FLOW", "SSLAM_CHOL_ORDER", "SSLAM_CHOL_USTAGE", "SSLAM_CHOL_RIGHT", "SSLAM_CHOL_SMALL_COLS",
                                       "SSLAM_CHOL_GROUP_CAP", "SSLAM_CHOL_MID_WIDTH", "SSLAM_FLOW_DEFER", "SSLAM_FLOW_LMSTEP"};
        for (const char* v : legacy)
          if (getenv(v)) fprintf(stderr, "[sslam] %s is no longer read: plan options go through SSLAM_CHOL_OPTS=\"key=value,...\" (chol_plan.hpp CholOpts)\n", v);
      }
    }
    const char* e = getenv("SSLAM_CHOL_OPTS");
    if (!e) return;
    std::string str(e);
    size_t pos = 0;
    while (pos < str.size()) {
      size_t end = str.find(',', pos);
      if (end == std::string::npos) end = str.size();
      const std::string kv = str.substr(pos, end - pos);
      pos = end + 1;
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) continue;
      const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
      const int iv = atoi(v.c_str());
      if (k == "cap_leaf") cap_leaf = iv; else if (k == "cap_mid") cap_mid = iv; else if (k == "cap_tail") cap_tail = iv;
      else if (k == "max_blocks") max_blocks = iv; else if (k == "tail_width") tail_width = iv; else if (k == "mid_width") mid_width = iv;
      else if (k == "nt_leaf") { nt_leaf = iv; nt_leaf_set = true; } else if (k == "nt_mid") nt_mid = iv; else if (k == "nt_ftail") nt_ftail = iv; else if (k == "nt_bleaf") nt_bleaf = iv; else if (k == "nt_bmid") nt_bmid = iv; else if (k == "nt_btail") nt_btail = iv; else if (k == "nt_tail") nt_tail = iv;
      else if (k == "min_chunk") min_chunk = std::max(1, iv); else if (k == "split_min") split_min = std::max(2, iv);
      else if (k == "pcap_leaf") pcap_leaf = iv; else if (k == "pcap_mid") pcap_mid = iv; else if (k == "pcap_tail") pcap_tail = iv;
      else if (k == "group_cap") group_cap = iv; else if (k == "group_blocks") group_blocks = iv; else if (k == "ustage") ustage = iv;
      else if (k == "order_bits_max") order_bits_max = iv; else if (k == "flow") flow = iv; else if (k == "front") front = iv; else if (k == "dump") dump = iv != 0;
      else if (k == "order") order = (v == "mindeg" || v == "0") ? 0 : 1;
      else if (k == "order_mul") order_mul = atof(v.c_str()); else if (k == "order_add") order_add = iv;
      else fprintf(stderr, "[sslam] SSLAM_CHOL_OPTS: unknown key '%s'\n", k.c_str());
    }
  }
  bool nt_leaf_set = false;   // nt_leaf came from SSLAM_CHOL_OPTS (the single-launch solve leaves it alone then)
};

// The options a plan is really built with: the workgroup sizes the kernels exist for, and the small-batch regime -- batches < 8
// (latency-bound: the orchestrator's graph, a single large graph) run the dependency-driven single launch (k_chol_flow) with every piece
// cut for the tail's workgroup size and no mid class.  ONE place for chol_plan_build and the plan introspection of the CPU tests (round-5
// ADVICE: the tests pinned plans cut for 64-thread pieces while the tick factored 512-thread ones).  Returns whether the plan is meant
// for the single launch.
// The throughput regime (groups of leaf pieces on 128 threads, mid class, front tables) is for batches that fill the chip with independent
// pieces: 32 graphs of any size, or two and more LARGE ones (>= 8000 block rows in the batch; round 6: 2 / 3 / 4 / 16 5000-pose graphs
// 1.96 k -> 2.04 k / 2.71 k -> 2.93 k / 3.39 k -> 3.81 k / 7.75 k -> 9.60 k LM iterations/s against the plans small batches took before; one such
// graph is as fast either way and keeps its 512-thread pieces).
inline bool chol_throughput_regime(int B, int block_rows) { return B >= 32 || (B >= 2 && block_rows >= 8000); }
inline bool chol_opts_normalise(CholOpts& opt, int B, int block_rows) {
  if (opt.nt_tail != 1024) opt.nt_tail = 512;
  if (opt.nt_leaf != -1 && opt.nt_leaf != 128 && opt.nt_leaf != 256 && opt.nt_leaf != 512 && opt.nt_leaf != 1024) opt.nt_leaf = 64;   // -1: by batch size (chol_symbolic)
  if (opt.nt_mid != 256 && opt.nt_mid != 512) opt.nt_mid = 128;
  const bool want_flow = opt.flow != 0 && B < 8 && !chol_throughput_regime(B, block_rows) && opt.nt_tail == 512 && opt.group_cap <= 0 && !opt.nt_leaf_set;
  if (want_flow) { opt.nt_leaf = opt.nt_tail; opt.mid_width = 0; }
  return want_flow;
}

struct CholHost {
  int ncol = 0, nlevels = 0, dim = 0, B = 0, npiece = 0;
  int64_t lnz = 0;           // doubles in Lval
  int64_t unz = 0;           // doubles in Uval (update matrices handed between pieces)
  std::vector<ColMeta> col; std::vector<BlkMeta> blk; std::vector<UpdMeta> upd; std::vector<ItemMeta> item; std::vector<MbMeta> mb;
  std::vector<ILevel> ilv; std::vector<PieceMeta> piece;
  std::vector<AsmSrc> asrc, usrc; std::vector<FwdMeta> fwd; std::vector<UItem> uitem; std::vector<UMb> umb;
  std::vector<RCol> rcol; std::vector<UpdMeta> rupd;   // right-looking update lists of the tail pieces, [ncol] and per piece
  std::vector<int> lvl_ptr, lvl_cols;       // column levels of the elimination tree (multi right-hand-side solves)
  std::vector<int> plv_ptr, plv_pieces;     // pieces grouped by depth (one launch each)
  std::vector<PieceMeta> lpiece;            // piece records in launch order: plv_pieces then tail_pieces (one dependent load less per workgroup)
  std::vector<int> tail_ptr, tail_pieces;   // per graph: its tail pieces in elimination order
  std::vector<int> plv_lds_f, plv_lds_b;    // LDS doubles per launch (factor / backward)
  std::vector<int> plv_nt, plv_cls;         // workgroup size (nt_leaf | nt_mid) and class (0 leaf | 1 mid) per launch
  int tail_lds_f = 0, tail_lds_b = 0;
  int nt_leaf = 64, nt_mid = 128, nt_tail = 512, nt_ftail = 256, nt_bleaf = 0, nt_bmid = 0, nt_btail = 0, ustage = 0;
  // front tables (front_plan.hpp); empty when the plan has none (front_why says why)
  bool front = false;
  std::string front_why;
  std::vector<uint32_t> fblob;
  std::vector<FrontGrp> fgrp, lfgrp;        // by piece id / in launch order (like lpiece)
  std::vector<int> plv_lds_ff;              // LDS doubles of the front factor kernel per launch
  int tail_lds_ff = 0;
  int64_t funz = 0;                         // doubles of update matrices in the front layout
  std::string error;
};

namespace chol_detail {

struct GraphSym {
  std::vector<int> order;                  // elimination order (local node ids)
  std::vector<int> cs_start, cs_len, cs_idx;   // per node: higher-ordered neighbours at elimination time, cs_idx[cs_start[v] .. + cs_len[v]) ascending
};

// minimum degree with explicit fill (the block graphs here have ~1e4 nodes and fill ~1.7x)
inline void min_degree(int n, std::vector<std::vector<int>>& adj, GraphSym& out) {
  std::vector<char> done(n, 0);
  using Item = std::pair<int, int>;  // (degree, node); lazy deletion
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  for (int v = 0; v < n; ++v) { std::sort(adj[v].begin(), adj[v].end()); pq.push({(int)adj[v].size(), v}); }
  out.order.clear(); out.order.reserve(n);
  out.cs_start.assign(n, 0); out.cs_len.assign(n, 0); out.cs_idx.clear();
  std::vector<int> merged;
  while (!pq.empty()) {
    const Item it = pq.top(); pq.pop();
    const int v = it.second;
    if (done[v] || it.first != (int)adj[v].size()) continue;
    done[v] = 1;
    out.order.push_back(v);
    std::vector<int>& nb = adj[v];
    out.cs_start[v] = (int)out.cs_idx.size(); out.cs_len[v] = (int)nb.size(); out.cs_idx.insert(out.cs_idx.end(), nb.begin(), nb.end());
    for (int u : nb) {
      std::vector<int>& au = adj[u];
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

// Multiple minimum degree over independent sets.  Plain minimum degree breaks ties by index, and on the pose chain of a SLAM graph
// (every interior pose has the same degree) that eliminates the chain from one end: the elimination tree of a 5000-pose graph was 151-189
// columns deep, and every level is a dependent step of the factorisation AND of both triangular solves.  Here a round takes every node
// whose degree is within a slack of the current minimum, keeps a maximal independent set of them (ascending degree, then index) and
// eliminates the whole set: its members do not touch each other, so they share a level of the tree.  A chain halves per round (the
// cyclic-reduction order), the tree of the same graph is 58-65 columns deep for +6 % fill; CSparse's AMD in the reference
// (graph_slam.cpp:67-73 -> LinearSolverCSparse) is likewise just *a* fill-reducing order: any order solves the same system.
inline void multi_min_degree(int n, std::vector<std::vector<int>>& adj, GraphSym& out, double mul, int add) {
  std::vector<char> alive(n, 1), blocked(n, 0);
  std::vector<int> live(n), cand, picked, merged;
  for (int v = 0; v < n; ++v) { std::sort(adj[v].begin(), adj[v].end()); live[v] = v; }
  out.order.clear(); out.order.reserve(n);
  out.cs_start.assign(n, 0); out.cs_len.assign(n, 0); out.cs_idx.clear();
  while (!live.empty()) {
    int mind = n;
    for (int v : live) mind = std::min(mind, (int)adj[v].size());
    const int lim = (int)(mul * mind) + add;
    cand.clear();
    for (int v : live) if ((int)adj[v].size() <= lim) cand.push_back(v);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
    picked.clear();
    for (int v : cand) {
      if (blocked[v]) continue;
      picked.push_back(v);
      for (int u : adj[v]) blocked[u] = 1;
    }
    for (int v : picked) for (int u : adj[v]) blocked[u] = 0;
    for (int v : picked) {
      alive[v] = 0;
      out.order.push_back(v);
      std::vector<int>& nb = adj[v];
      out.cs_start[v] = (int)out.cs_idx.size(); out.cs_len[v] = (int)nb.size(); out.cs_idx.insert(out.cs_idx.end(), nb.begin(), nb.end());
      for (int u : nb) {
        std::vector<int>& au = adj[u];
        merged.clear();
        merged.reserve(au.size() + nb.size());
        std::set_union(au.begin(), au.end(), nb.begin(), nb.end(), std::back_inserter(merged));
        au.clear();
        for (int w : merged) if (w != u && w != v) au.push_back(w);
      }
      std::vector<int>().swap(adj[v]);
    }
    size_t k = 0;
    for (int v : live) if (alive[v]) live[k++] = v;
    live.resize(k);
  }
}

// The same algorithm on adjacency BITSETS, for the graphs the orchestrator's tick re-plans every time it runs (a few hundred nodes: a row is
// a handful of words, the fill of an elimination is one OR per neighbour, a node's structure is read off its row in ascending order).
// Produces exactly the order and the column structures of multi_min_degree (same candidates, same tie breaks); 0.44 -> 0.09 ms of the
// 2.1 ms host phase at 436 keyframes.
template <class Neighbours>   // neighbours(v, f): calls f(u) for every neighbour u of node v
inline void multi_min_degree_bits(int n, Neighbours&& neighbours, GraphSym& out, double mul, int add) {
  const int W = (n + 63) / 64;
  std::vector<uint64_t> bits((size_t)n * W, 0);
  std::vector<int> deg(n, 0);
  auto row = [&](int v) { return bits.data() + (size_t)v * W; };
  for (int v = 0; v < n; ++v) neighbours(v, [&](int u) { row(v)[u >> 6] |= 1ull << (u & 63); });
  for (int v = 0; v < n; ++v) { int d = 0; for (int w = 0; w < W; ++w) d += __builtin_popcountll(row(v)[w]); deg[v] = d; }
  std::vector<char> alive(n, 1), blocked(n, 0);
  std::vector<int> live(n), cand, picked, cnt;
  for (int v = 0; v < n; ++v) live[v] = v;
  out.order.clear(); out.order.reserve(n);
  out.cs_start.assign(n, 0); out.cs_len.assign(n, 0); out.cs_idx.clear();
  auto for_bits = [&](const uint64_t* r, auto&& f) {
    for (int w = 0; w < W; ++w) { uint64_t m = r[w]; while (m) { const int b = __builtin_ctzll(m); m &= m - 1; f(w * 64 + b); } }
  };
  while (!live.empty()) {
    int mind = n;
    for (int v : live) mind = std::min(mind, deg[v]);
    const int lim = (int)(mul * mind) + add;
    // candidates by (degree, index): live ascends, a counting sort by degree keeps the index order inside a degree
    cnt.assign(lim - mind + 2, 0);
    for (int v : live) if (deg[v] <= lim) cnt[deg[v] - mind + 1]++;
    for (int d = 0; d + 1 < (int)cnt.size(); ++d) cnt[d + 1] += cnt[d];
    cand.resize(cnt.back());
    for (int v : live) if (deg[v] <= lim) cand[cnt[deg[v] - mind]++] = v;
    picked.clear();
    for (int v : cand) {
      if (blocked[v]) continue;
      picked.push_back(v);
      for_bits(row(v), [&](int u) { blocked[u] = 1; });
    }
    for (int v : picked) for_bits(row(v), [&](int u) { blocked[u] = 0; });
    for (int v : picked) {
      alive[v] = 0;
      out.order.push_back(v);
      uint64_t* rv = row(v);
      const int cs0 = (int)out.cs_idx.size();
      out.cs_start[v] = cs0;
      for_bits(rv, [&](int u) { out.cs_idx.push_back(u); });
      const int cs1 = (int)out.cs_idx.size();
      out.cs_len[v] = cs1 - cs0;
      for (int ci = cs0; ci < cs1; ++ci) {
        const int u = out.cs_idx[ci];
        uint64_t* ru = row(u);
        int d = 0;
        for (int w = 0; w < W; ++w) ru[w] |= rv[w];
        ru[u >> 6] &= ~(1ull << (u & 63));
        ru[v >> 6] &= ~(1ull << (v & 63));
        for (int w = 0; w < W; ++w) d += __builtin_popcountll(ru[w]);
        deg[u] = d;
      }
      for (int w = 0; w < W; ++w) rv[w] = 0;
      deg[v] = 0;
    }
    size_t k = 0;
    for (int v : live) if (alive[v]) live[k++] = v;
    live.resize(k);
  }
}

// Greedy bottom-up cut of an elimination tree (columns 0..n-1 in elimination order, parent[s] > s or -1) into pieces:
// a column joins the still-open pieces of its children, largest first, while the caps hold; whatever does not fit is
// closed.  fixed[s] >= 0 pins column s to an existing piece id (treated as closed).  cls (optional): class of every column; a column only
// joins pieces of its own class and the cap is caps[class] (classes ascend towards the root).  Returns the piece id per column
// (ids of new pieces start at first_new_id) and the number of ids used.
inline int cut_pieces(int n, const std::vector<int>& parent, const std::vector<int>& colsz, const std::vector<int>& colnb,
                      const std::vector<int>& fixed, int first_new_id, int cap_all, int max_blocks, std::vector<int>& pc,
                      const std::vector<int>* cls = nullptr, const int* caps = nullptr) {
  pc.assign(n, -1);
  // children lists in CSR form (ascending inside a list, like the per-column vectors they replace); the columns of a piece as a linked
  // list (only ever walked to relabel them); scratch allocated once
  std::vector<int> kptr(n + 1, 0), kidx(n);
  for (int s = 0; s < n; ++s) if (parent[s] >= 0) kptr[parent[s] + 1]++;
  for (int s = 0; s < n; ++s) kptr[s + 1] += kptr[s];
  {
    std::vector<int> cur(kptr.begin(), kptr.end() - 1);
    for (int s = 0; s < n; ++s) if (parent[s] >= 0) kidx[cur[parent[s]]++] = s;
  }
  std::vector<int> mhead, mtail, mnext(n, -1);   // per new piece (index = id - first_new_id): first / last column; next column of a column
  std::vector<int> psize, pblk;
  std::vector<char> open;
  std::vector<int> il(n, 0);               // level of a column inside its piece
  std::vector<int> cand, take;
  for (int s = 0; s < n; ++s) {
    if (fixed[s] >= 0) { pc[s] = fixed[s]; continue; }
    cand.clear();                          // open pieces of the children
    for (int kq = kptr[s]; kq < kptr[s + 1]; ++kq) {
      const int c = kidx[kq];
      const int p = pc[c];
      if (fixed[c] >= 0) continue;
      if (cls && (*cls)[c] != (*cls)[s]) continue;
      if (open[p - first_new_id] && std::find(cand.begin(), cand.end(), p) == cand.end()) cand.push_back(p);
    }
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
      const int sa = psize[a - first_new_id], sb = psize[b - first_new_id];
      return sa != sb ? sa > sb : a < b;
    });
    int size = colsz[s], nb = colnb[s], lev = 0;
    const int cap = cls ? caps[(*cls)[s]] : cap_all;
    take.clear();
    for (int p : cand) {
      const int q = p - first_new_id;
      int l = 0;
      for (int kq = kptr[s]; kq < kptr[s + 1]; ++kq) if (pc[kidx[kq]] == p) l = std::max(l, il[kidx[kq]] + 1);
      if (size + psize[q] <= cap && nb + pblk[q] <= max_blocks && std::max(lev, l) < kMaxILevels) {
        size += psize[q]; nb += pblk[q]; lev = std::max(lev, l); take.push_back(p);
      }
    }
    for (int p : cand) open[p - first_new_id] = 0;   // merged or closed: either way no longer a candidate
    int id;
    if (take.empty()) {
      id = first_new_id + (int)mhead.size();
      mhead.push_back(-1); mtail.push_back(-1); psize.push_back(0); pblk.push_back(0); open.push_back(1);
    } else {
      id = take[0];                                   // the largest child keeps its id; the others are relabelled into it
      const int qd = id - first_new_id;
      for (size_t k = 1; k < take.size(); ++k) {
        const int qs = take[k] - first_new_id;
        for (int c = mhead[qs]; c >= 0; c = mnext[c]) pc[c] = id;
        if (mhead[qs] >= 0) { if (mtail[qd] >= 0) mnext[mtail[qd]] = mhead[qs]; else mhead[qd] = mhead[qs]; mtail[qd] = mtail[qs]; }
        mhead[qs] = mtail[qs] = -1;
      }
    }
    const int q = id - first_new_id;
    if (mtail[q] >= 0) mnext[mtail[q]] = s; else mhead[q] = s;
    mtail[q] = s;
    pc[s] = id; psize[q] = size; pblk[q] = nb; open[q] = 1; il[s] = lev;
  }
  return first_new_id + (int)mhead.size();
}

}  // namespace chol_detail

// Symbolic factorisation + piece plan of a whole batch.  Returns 0, or -1 with out.error set.
inline int chol_symbolic(const SymIn& in, CholOpts opt, CholHost& out) {
  using namespace chol_detail;
  SSLAM_PT_INIT
  const int nPr = in.nPr, nLr = in.nLr, nrow = nPr + nLr, B = in.B;
  if (opt.tail_width < 0) opt.tail_width = 2;
  const bool thr = chol_throughput_regime(B, nrow);
  if (opt.ustage < 0) opt.ustage = thr ? 0 : 1;
  if (opt.mid_width < 0) opt.mid_width = thr ? 60 : 0;
  if (opt.cap_leaf < 0) opt.cap_leaf = thr ? 700 : 900;
  if (opt.group_cap < 0) opt.group_cap = thr ? (B >= 32 ? 2800 : 1400) : 0;   // (a few large graphs: smaller groups, more of them)
  if (opt.nt_leaf < 0) opt.nt_leaf = thr ? 128 : 64;
  if (opt.order < 0) opt.order = 1;
  if (opt.order_mul < 0 || opt.order_add < 0) { opt.order_mul = B >= 32 ? 1.5 : 2.0; opt.order_add = B >= 32 ? 2 : 4; }   // (a few large graphs keep the shallowest tree)
  if (opt.front < 0) opt.front = thr ? 1 : 0;
  out = CholHost();
  out.B = B; out.nt_leaf = opt.nt_leaf; out.nt_mid = opt.nt_mid; out.nt_tail = opt.nt_tail; out.nt_ftail = opt.nt_ftail; out.nt_bleaf = opt.nt_bleaf; out.nt_bmid = opt.nt_bmid; out.nt_btail = opt.nt_btail; out.ustage = opt.ustage;
  auto row_dim = [&](int r) { return r < nPr ? 6 : 3; };
  auto row_xoff = [&](int r) { return r < nPr ? 6 * r : 6 * nPr + 3 * (r - nPr); };
  // adjacency of the block graph in CSR form (rows: pose rows, then landmark rows), every edge with the offset of its block in H: three
  // flat arrays instead of a hash map and a vector per row (the plan is rebuilt at every tick of the orchestrator)
  std::vector<int> aptr(nrow + 1, 0), aidx, aoff;
  {
    for (auto& pr : in.ppoff) { aptr[pr.first + 1]++; aptr[pr.second + 1]++; }
    for (auto& pr : in.plblk) { aptr[pr.first + 1]++; aptr[nPr + pr.second + 1]++; }
    for (auto& pr : in.llblk) { aptr[nPr + pr.first + 1]++; aptr[nPr + pr.second + 1]++; }
    for (int r = 0; r < nrow; ++r) aptr[r + 1] += aptr[r];
    aidx.resize(aptr[nrow]); aoff.resize(aptr[nrow]);
    std::vector<int> cur(aptr.begin(), aptr.end() - 1);
    auto edge = [&](int a, int c, int64_t off) { aidx[cur[a]] = c; aoff[cur[a]++] = (int)off; aidx[cur[c]] = a; aoff[cur[c]++] = (int)off; };
    for (size_t i = 0; i < in.ppoff.size(); ++i) edge(in.ppoff[i].first, in.ppoff[i].second, in.hpp_off_base + (int64_t)i * 36);
    for (size_t i = 0; i < in.plblk.size(); ++i) edge(in.plblk[i].first, nPr + in.plblk[i].second, in.hpl_base + (int64_t)i * 18);
    for (size_t i = 0; i < in.llblk.size(); ++i) edge(nPr + in.llblk[i].first, nPr + in.llblk[i].second, in.hll_off_base + (int64_t)i * 9);
  }
  auto h_offset = [&](int a, int c) -> int {   // offset of block (a, c) of H, or -1 (fill)
    for (int q = aptr[a]; q < aptr[a + 1]; ++q) if (aidx[q] == c) return aoff[q];
    return -1;
  };

  // ---- per graph: ordering, elimination tree, pieces, final (piece-contiguous) elimination order ------------------
  std::vector<int> col_row, col_graph, col_piece, col_comp, col_tail;   // by final column id; piece = execution group, comp = connected piece, tail = class (0 leaf, 1 mid, 2 tail)
  std::vector<int> row_col(nrow, -1);
  std::vector<int> cr_ptr{0}, cr_idx;                         // per column: rows of the off-diagonal blocks (CSR)
  int npiece = 0, ncomp = 0;
  for (int g = 0; g < B; ++g) {
    const SymGraph& sg = in.seg[g];
    const int n = sg.nprow + sg.nlrow;
    auto loc2row = [&](int v) { return v < sg.nprow ? sg.prow0 + v : nPr + sg.lrow0 + (v - sg.nprow); };
    auto row2loc = [&](int r) { return r < nPr ? r - sg.prow0 : sg.nprow + (r - nPr - sg.lrow0); };
    GraphSym S;
    if (opt.order == 1 && n <= opt.order_bits_max) {
      multi_min_degree_bits(n, [&](int v, auto&& f) { const int r = loc2row(v); for (int q = aptr[r]; q < aptr[r + 1]; ++q) f(row2loc(aidx[q])); }, S, opt.order_mul, opt.order_add);
    } else {
      std::vector<std::vector<int>> ladj(n);
      for (int v = 0; v < n; ++v) {
        const int r = loc2row(v);
        ladj[v].reserve(aptr[r + 1] - aptr[r]);
        for (int q = aptr[r]; q < aptr[r + 1]; ++q) ladj[v].push_back(row2loc(aidx[q]));
      }
      if (opt.order == 1) multi_min_degree(n, ladj, S, opt.order_mul, opt.order_add);
      else min_degree(n, ladj, S);
    }
    std::vector<int> pos(n);
    for (int s = 0; s < n; ++s) pos[S.order[s]] = s;
    std::vector<int> parent(n, -1), colsz(n), colnb(n);
    for (int s = 0; s < n; ++s) {
      const int v = S.order[s];
      const int d = row_dim(loc2row(v));
      int rows = 0, par = n;
      for (int q = S.cs_start[v]; q < S.cs_start[v] + S.cs_len[v]; ++q) { const int w = S.cs_idx[q]; rows += row_dim(loc2row(w)); par = std::min(par, pos[w]); }
      parent[s] = par == n ? -1 : par;
      colsz[s] = d * d + d * rows;
      colnb[s] = 1 + S.cs_len[v];
    }
    // pass 1: leaf-sized pieces everywhere; depth of every piece in the piece tree
    std::vector<int> pc, none(n, -1);
    int np1 = cut_pieces(n, parent, colsz, colnb, none, 0, opt.cap_leaf, opt.max_blocks, pc);
    std::vector<int> plev(np1, 0);
    for (int s = 0; s < n; ++s)
      if (parent[s] >= 0 && pc[parent[s]] != pc[s]) plev[pc[parent[s]]] = std::max(plev[pc[parent[s]]], plev[pc[s]] + 1);
    // (a piece's depth must cover chains through several columns: iterate in elimination order, children first)
    {
      std::fill(plev.begin(), plev.end(), 0);
      std::vector<int> colplev(n, 0);   // depth of the piece as known when the column is reached
      bool changed = true;
      while (changed) {
        changed = false;
        for (int s = 0; s < n; ++s)
          if (parent[s] >= 0 && pc[parent[s]] != pc[s] && plev[pc[parent[s]]] < plev[pc[s]] + 1) { plev[pc[parent[s]]] = plev[pc[s]] + 1; changed = true; }
      }
    }
    int nlev1 = 0;
    for (int p = 0; p < np1; ++p) nlev1 = std::max(nlev1, plev[p] + 1);
    std::vector<int> cnt(nlev1, 0);
    for (int p = 0; p < np1; ++p) cnt[plev[p]]++;
    int T = nlev1;
    if (opt.tail_width > 0) while (T > 0 && cnt[T - 1] <= opt.tail_width) --T;
    // pass 2: the columns above the bottom are cut again, by class: mid (depths with <= mid_width pieces of this graph) with the mid cap,
    // tail with the tail cap (fewer external-update phases on the chain).  Depths ascend towards the root, so do the classes.
    int M = T;
    if (opt.mid_width > 0) while (M > 0 && cnt[M - 1] <= opt.mid_width) --M;
    std::vector<int> fixed(n, -1), ccls(n, 0);
    bool any_upper = false;
    for (int s = 0; s < n; ++s) {
      const int d = plev[pc[s]];
      ccls[s] = d < M ? 0 : (d < T ? 1 : 2);
      if (ccls[s] == 0) fixed[s] = pc[s]; else any_upper = true;
    }
    std::vector<int> pc2 = pc;
    int np2 = np1;
    const int caps[3] = {opt.cap_leaf, opt.cap_mid, opt.cap_tail};
    if (any_upper) np2 = cut_pieces(n, parent, colsz, colnb, fixed, np1, opt.cap_tail, opt.max_blocks, pc2, &ccls, caps);
    std::vector<char> is_tail(np2, 0), pcls(np2, 0);
    for (int s = 0; s < n; ++s) { pcls[pc2[s]] = (char)ccls[s]; if (ccls[s] == 2) is_tail[pc2[s]] = 1; }
    // depth of every connected piece ("component") in the piece tree of this graph
    std::vector<int> proot(np2, -1);
    for (int s = 0; s < n; ++s) proot[pc2[s]] = std::max(proot[pc2[s]], s);
    std::vector<int> il(n, 0);
    for (int s = 0; s < n; ++s) if (parent[s] >= 0 && pc2[parent[s]] == pc2[s]) il[parent[s]] = std::max(il[parent[s]], il[s] + 1);
    std::vector<int> ids;
    for (int p = 0; p < np2; ++p) if (proot[p] >= 0) ids.push_back(p);
    std::sort(ids.begin(), ids.end(), [&](int a, int b) { return proot[a] < proot[b]; });
    std::vector<int> cpar(np2, -1), cplev(np2, 0), csize(np2, 0), cblk(np2, 0);
    for (int s = 0; s < n; ++s) {
      if (parent[s] >= 0 && pc2[parent[s]] != pc2[s]) cpar[pc2[s]] = pc2[parent[s]];
      csize[pc2[s]] += colsz[s]; cblk[pc2[s]] += colnb[s];
    }
    for (int c : ids) if (cpar[c] >= 0) cplev[cpar[c]] = std::max(cplev[cpar[c]], cplev[c] + 1);   // ascending root: children first
    // execution groups: the non-tail components by (depth, root), packed side by side up to the group caps (group_cap = 0: one
    // component per group); then the tail components in elimination order, one per group.  The tail is closed upwards, so every
    // group only depends on groups before it.
    std::vector<int> gorder;
    for (int c : ids) if (!is_tail[c]) gorder.push_back(c);
    std::stable_sort(gorder.begin(), gorder.end(), [&](int a, int b) { return cplev[a] < cplev[b]; });
    std::vector<int> grank(np2, -1);
    int ngroup = 0;
    {
      int gsz = 0, gnb = 0, glev = -1;
      for (int c : gorder) {
        const bool fits = opt.group_cap > 0 && glev == cplev[c] && pcls[c] == 0 && gsz + csize[c] <= opt.group_cap && gnb + cblk[c] <= opt.group_blocks;
        if (!fits) { ++ngroup; gsz = 0; gnb = 0; glev = cplev[c]; }
        gsz += csize[c]; gnb += cblk[c];
        grank[c] = ngroup - 1;
      }
      for (int c : ids) if (is_tail[c]) grank[c] = ngroup++;
    }
    std::vector<int> crank(np2, -1);
    for (size_t k = 0; k < ids.size(); ++k) crank[ids[k]] = (int)k;
    // final order: groups; inside a group by (level inside the component, component, position): one contiguous column range per level
    std::vector<int> perm(n);
    for (int s = 0; s < n; ++s) perm[s] = s;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) {
      if (grank[pc2[a]] != grank[pc2[b]]) return grank[pc2[a]] < grank[pc2[b]];
      if (il[a] != il[b]) return il[a] < il[b];
      if (pc2[a] != pc2[b]) return crank[pc2[a]] < crank[pc2[b]];
      return a < b;
    });
    const int c0 = (int)col_row.size();
    for (int k = 0; k < n; ++k) {
      const int s = perm[k];
      const int r = loc2row(S.order[s]);
      row_col[r] = c0 + k;
      col_row.push_back(r); col_graph.push_back(g);
      col_piece.push_back(npiece + grank[pc2[s]]); col_comp.push_back(ncomp + crank[pc2[s]]); col_tail.push_back(pcls[pc2[s]]);
      const int vs = S.order[s];
      for (int q = S.cs_start[vs]; q < S.cs_start[vs] + S.cs_len[vs]; ++q) cr_idx.push_back(loc2row(S.cs_idx[q]));
      cr_ptr.push_back((int)cr_idx.size());
    }
    npiece += ngroup;
    ncomp += (int)ids.size();
  }
  SSLAM_PT("order+pieces")
  const int ncol = (int)col_row.size();
  out.ncol = ncol; out.npiece = npiece; out.dim = 6 * nPr + 3 * nLr;

  // ---- blocks of every column, sorted by elimination position of the row ----------------------------------------------
  std::vector<int> bp(ncol + 1, 0), boff, brow, bsrc, col_xoff(ncol), col_yoff(ncol), col_dim(ncol);
  std::vector<unsigned char> bfmt;
  int64_t lnz = 0;
  {
    int y = 0;
    std::vector<int> rows_c;
    for (int j = 0; j < ncol; ++j) {
      const int rj = col_row[j], dj = row_dim(rj);
      col_xoff[j] = row_xoff(rj); col_dim[j] = dj; col_yoff[j] = y; y += dj;
      rows_c.clear();
      for (int q = cr_ptr[j]; q < cr_ptr[j + 1]; ++q) rows_c.push_back(row_col[cr_idx[q]]);
      std::sort(rows_c.begin(), rows_c.end());
      bp[j] = (int)boff.size();
      boff.push_back((int)lnz); brow.push_back(j);
      bsrc.push_back(rj < nPr ? rj * 36 : (int)(in.hll_base + (int64_t)(rj - nPr) * 9)); bfmt.push_back(0);
      lnz += blk_doubles(dj, dj);
      for (int i : rows_c) {
        if (i <= j) { out.error = "symbolic factorisation inconsistent (row not below its column)"; return -1; }
        const int ri = col_row[i], di = row_dim(ri);
        boff.push_back((int)lnz); brow.push_back(i);
        const int a = std::min(ri, rj), c = std::max(ri, rj);
        const int ho = h_offset(a, c);
        if (ho < 0) { bsrc.push_back(-1); bfmt.push_back(0); }
        else { bsrc.push_back(ho); bfmt.push_back(ri == a ? 0 : 1); }   // stored [min][max]; we need [i][j]
        lnz += blk_doubles(di, dj);
      }
      if (lnz >= ((int64_t)1 << 31) - 4096) { out.error = "Cholesky factor too large for int32 offsets"; return -1; }
    }
  }
  bp[ncol] = (int)boff.size();
  const int nblk = (int)boff.size();
  // block (row i, column j), or -1: the rows of a column ascend (diagonal first)
  auto find_blk = [&](int j, int i) -> int {
    const int* lo = brow.data() + bp[j];
    const int* hi = brow.data() + bp[j + 1];
    const int* it = std::lower_bound(lo, hi, i);
    return (it != hi && *it == i) ? (int)(it - brow.data()) : -1;
  };
  out.lnz = lnz;
  // flat layout inside a piece: blocks sorted by size class (stable in block order)
  std::vector<int> piece_base, piece_size, piece_n36, piece_n18;
  {
    int j = 0;
    while (j < ncol) {
      int j1 = j;
      while (j1 < ncol && col_piece[j1] == col_piece[j]) ++j1;
      const int base = boff[bp[j]];
      int cnt[3] = {0, 0, 0};
      auto cls = [&](int t, int jj) { const int d = blk_doubles(col_dim[brow[t]], col_dim[jj]); return d == 36 ? 0 : (d == 18 ? 1 : 2); };
      for (int jj = j; jj < j1; ++jj)
        for (int t = bp[jj]; t < bp[jj + 1]; ++t) cnt[cls(t, jj)]++;
      const int size[3] = {36, 18, 10};
      int start[3] = {base, base + 36 * cnt[0], base + 36 * cnt[0] + 18 * cnt[1]};
      int idx[3] = {0, 0, 0};
      for (int jj = j; jj < j1; ++jj)
        for (int t = bp[jj]; t < bp[jj + 1]; ++t) { const int c = cls(t, jj); boff[t] = start[c] + size[c] * idx[c]++; }
      if ((int)piece_base.size() <= col_piece[j]) { piece_base.resize(col_piece[j] + 1, 0); piece_size.resize(col_piece[j] + 1, 0); piece_n36.resize(col_piece[j] + 1, 0); piece_n18.resize(col_piece[j] + 1, 0); }
      piece_base[col_piece[j]] = base; piece_size[col_piece[j]] = 36 * cnt[0] + 18 * cnt[1] + 10 * cnt[2];
      piece_n36[col_piece[j]] = cnt[0]; piece_n18[col_piece[j]] = cnt[1];
      j = j1;
    }
  }
  SSLAM_PT("blocks")
  // ---- levels of the block elimination tree (multi right-hand-side solves) and inside the pieces --------------------------
  std::vector<int> level(ncol, 0), col_il(ncol, 0);
  int nlev = 0;
  for (int j = 0; j < ncol; ++j) {
    if (bp[j + 1] - bp[j] > 1) {
      const int par = brow[bp[j] + 1];
      level[par] = std::max(level[par], level[j] + 1);
      if (col_comp[par] == col_comp[j]) col_il[par] = std::max(col_il[par], col_il[j] + 1);
    }
    nlev = std::max(nlev, level[j] + 1);
  }
  out.nlevels = nlev;
  out.lvl_ptr.assign(nlev + 1, 0);
  for (int j = 0; j < ncol; ++j) out.lvl_ptr[level[j] + 1]++;
  for (int l = 0; l < nlev; ++l) out.lvl_ptr[l + 1] += out.lvl_ptr[l];
  out.lvl_cols.resize(ncol);
  {
    std::vector<int> cursor(out.lvl_ptr.begin(), out.lvl_ptr.end() - 1);
    for (int j = 0; j < ncol; ++j) out.lvl_cols[cursor[level[j]]++] = j;
  }
  // ---- pieces: column / block ranges (contiguous by construction), parent piece, depth ----------------------------------------
  out.piece.assign(npiece, PieceMeta{});
  for (auto& pm0 : out.piece) pm0.pad4 = -1;
  std::vector<char> piece_tail(npiece, 0), piece_cls(npiece, 0);
  std::vector<int> comp_parent(ncomp, -1), comp_dest(ncomp, -1);   // parent component; the group that holds it
  for (int j = 0; j < ncol; ++j) {
    PieceMeta& pm = out.piece[col_piece[j]];
    if (pm.nc == 0) { pm.graph = col_graph[j]; pm.c0 = j; pm.b0 = bp[j]; pm.lbase = piece_base[col_piece[j]]; pm.y0 = col_yoff[j];
                      pm.lsize = piece_size[col_piece[j]]; pm.n36 = piece_n36[col_piece[j]]; pm.n18 = piece_n18[col_piece[j]]; }
    if (j != pm.c0 + pm.nc) { out.error = "piece columns are not contiguous"; return -1; }
    pm.nc++;
    pm.nb = bp[j + 1] - pm.b0;
    pm.ysize = col_yoff[j] + col_dim[j] - pm.y0;
    piece_tail[col_piece[j]] = (char)(col_tail[j] == 2); piece_cls[col_piece[j]] = (char)col_tail[j]; pm.pad5 = col_tail[j];
    if (bp[j + 1] - bp[j] > 1) {
      const int par = brow[bp[j] + 1];
      if (col_comp[par] != col_comp[j]) {
        if (comp_parent[col_comp[j]] >= 0 && comp_parent[col_comp[j]] != col_comp[par]) { out.error = "a connected piece has two parents"; return -1; }
        if (col_piece[par] <= col_piece[j]) { out.error = "groups are not in elimination order"; return -1; }
        comp_parent[col_comp[j]] = col_comp[par];
        comp_dest[col_comp[j]] = col_piece[par];
        // the piece its update matrix goes to (pad4; -1: a root).  A group of several components may hang below several pieces (group_cap > 0):
        // -2 then, and the dependency-driven kernel (k_chol_flow) is not used for such a plan
        int& pp4 = out.piece[col_piece[j]].pad4;
        pp4 = (pp4 == -1 || pp4 == col_piece[par]) ? col_piece[par] : -2;
      }
    }
  }
  std::vector<int> plev(npiece, 0);
  for (int j = 0; j < ncol; ++j)   // ascending columns = ascending groups: a group's depth is final before any of its parents' columns is visited
    if (bp[j + 1] - bp[j] > 1) {
      const int pp = col_piece[brow[bp[j] + 1]], pj = col_piece[j];
      if (pp != pj) plev[pp] = std::max(plev[pp], plev[pj] + 1);
    }
  int nplv = 0;
  for (int p = 0; p < npiece; ++p) if (!piece_tail[p]) nplv = std::max(nplv, plev[p] + 1);
  out.plv_ptr.assign(nplv + 1, 0);
  for (int p = 0; p < npiece; ++p) if (!piece_tail[p]) out.plv_ptr[plev[p] + 1]++;
  for (int l = 0; l < nplv; ++l) out.plv_ptr[l + 1] += out.plv_ptr[l];
  out.plv_pieces.resize(out.plv_ptr[nplv]);
  {
    std::vector<int> cursor(out.plv_ptr.begin(), out.plv_ptr.end() - 1);
    for (int p = 0; p < npiece; ++p) if (!piece_tail[p]) out.plv_pieces[cursor[plev[p]]++] = p;
  }
  out.tail_ptr.assign(B + 1, 0);
  for (int p = 0; p < npiece; ++p) if (piece_tail[p]) out.tail_ptr[out.piece[p].graph + 1]++;
  for (int g = 0; g < B; ++g) out.tail_ptr[g + 1] += out.tail_ptr[g];
  out.tail_pieces.resize(out.tail_ptr[B]);
  {
    std::vector<int> cursor(out.tail_ptr.begin(), out.tail_ptr.end() - 1);
    for (int p = 0; p < npiece; ++p) if (piece_tail[p]) out.tail_pieces[cursor[out.piece[p].graph]++] = p;   // ascending id = elimination order
  }
  // ---- columns, blocks, rows (forward lists) ------------------------------------------------------------------------------------
  out.col.assign(ncol, ColMeta{});
  out.blk.assign(nblk, BlkMeta{});
  std::vector<int> block_col(nblk), col_nbi(ncol, 0);
  {
    std::vector<int> rowcnt(ncol + 1, 0);
    for (int j = 0; j < ncol; ++j) for (int t = bp[j] + 1; t < bp[j + 1]; ++t) rowcnt[brow[t] + 1]++;
    for (int j = 0; j < ncol; ++j) rowcnt[j + 1] += rowcnt[j];
    out.fwd.resize(rowcnt[ncol]);
    std::vector<int> cursor(rowcnt.begin(), rowcnt.end() - 1);
    for (int j = 0; j < ncol; ++j) {
      const int pj = col_piece[j];
      int nbi = 0;
      for (int t = bp[j]; t < bp[j + 1]; ++t) if (col_piece[brow[t]] == pj) ++nbi;   // rows sorted by position: in-piece rows come first
      for (int t = bp[j]; t < bp[j] + nbi; ++t) if (col_piece[brow[t]] != pj) { out.error = "in-piece rows of a column are not a prefix"; return -1; }
      col_nbi[j] = nbi;
      out.col[j] = ColMeta{col_xoff[j], col_yoff[j], col_dim[j], col_graph[j], bp[j], bp[j + 1] - bp[j], nbi, boff[bp[j]], rowcnt[j], rowcnt[j + 1], pj, col_il[j]};
      for (int t = bp[j]; t < bp[j + 1]; ++t) {
        const int i = brow[t];
        block_col[t] = j;
        BlkMeta& bm = out.blk[t];
        bm.off = boff[t]; bm.src = bsrc[t];
        bm.xoff_row = col_xoff[i]; bm.yoff_row = col_yoff[i]; bm.coldiag = boff[bp[j]]; bm.colyoff = col_yoff[j]; bm.as0 = 0;
        bm.info = col_dim[i] | (col_dim[j] << 4) | (bfmt[t] ? kBlkFmt : 0) | (t == bp[j] ? kBlkDiag : 0) | (col_piece[i] == pj ? kBlkRowIn : 0);
        if (t > bp[j]) out.fwd[cursor[i]++] = FwdMeta{boff[t] | (col_dim[j] == 6 ? (int)0x80000000u : 0), col_yoff[j]};   // columns ascend: row lists sorted by k
      }
    }
  }
  for (int p = 0; p < npiece; ++p) {   // slots of the row-in-piece blocks, in block order
    PieceMeta& pm = out.piece[p];
    int r = 0;
    for (int t = pm.b0; t < pm.b0 + pm.nb; ++t)
      if (out.blk[t].info & kBlkRowIn) {
        if (r >= 8192) { out.error = "piece with more than 8191 row-in-piece blocks"; return -1; }
        out.blk[t].info |= blk_irank_bits(r++);
      }
    pm.nint = r;
  }
  SSLAM_PT("levels+cols")
  // ---- piece by piece (elimination order: children before parents): internal updates, update matrix, assembly -------------------------
  struct URec { int a, b, uoff, uy, comp; };          // finished update-matrix block (row column-ids a >= b; uy: rhs part of a diagonal block) of component comp
  std::vector<std::vector<URec>> inbox(npiece);       // per group: the blocks its child components handed up (kept until consumed)
  std::vector<std::vector<int>> comp_R(ncomp);        // boundary rows of a component (column ids, ascending)
  out.upd.clear(); out.item.clear(); out.mb.clear(); out.ilv.clear(); out.asrc.clear(); out.usrc.clear(); out.uitem.clear(); out.umb.clear();
  out.rupd.clear(); out.rcol.assign(ncol, RCol{0, 0});
  bool right_ok = true;
  std::vector<int> piece_pmax(npiece, 0);   // most partial tiles any phase of the piece needs
  int64_t ucur = 0;
  // scratch of the piece loop, allocated once (the loop runs per tick of the orchestrator: no allocation per piece)
  struct UB { int comp, a, b; };
  struct OwnRec { int blk, ua, ubo, k; };
  struct SrcRec { int blk; AsmSrc s; };
  struct IU { int t, ua, ub2, k; };
  struct AsmRec { int t; AsmSrc s; };
  std::vector<UB> ub;
  std::vector<int> comps, ctab_base, ctab, rix, iu_ptr, as_ptr, own_ptr, src_ptr, cur1, cur2, cur3, cur4, bi0, bi1, order, uoffs;
  std::vector<OwnRec> own_flat, own_s;
  std::vector<SrcRec> src_flat;
  std::vector<IU> iu_flat, iu_s;
  std::vector<AsmRec> asm_flat;
  std::vector<AsmSrc> as_s, src_s;
  std::vector<int> lens, comp_uy;
  std::vector<UItem> uit_s;
  // smallest chunk >= lo for which the lists longer than the chunk are cut into <= pcap items in total (or chunk >= hi): the count falls
  // monotonically with the chunk, so bisection finds what the linear search found
  auto fit_chunk = [&](int lo, int hi, int pcap) {
    auto nonsole = [&](int chunk) { int s2 = 0; for (int n : lens) { const int k = (n + chunk - 1) / chunk; if (k > 1) s2 += k; } return s2; };
    if (lo >= hi || nonsole(lo) <= pcap) return lo;
    int a = lo, b = hi;   // nonsole(a) > pcap; b is accepted
    while (b - a > 1) { const int m = a + (b - a) / 2; if (nonsole(m) <= pcap) b = m; else a = m; }
    return b;
  };
  for (int p = 0; p < npiece; ++p) {
    PieceMeta& pm = out.piece[p];
    const int nt = piece_tail[p] ? opt.nt_tail : (piece_cls[p] == 1 ? opt.nt_mid : opt.nt_leaf);
    const int slots = nt / 4;
    const int pcap = piece_tail[p] ? opt.pcap_tail : (piece_cls[p] == 1 ? opt.pcap_mid : opt.pcap_leaf);   // partial tiles a phase may use (LDS: 336 B each)
    // boundary rows of every component of the group
    comps.clear();
    for (int j = pm.c0; j < pm.c0 + pm.nc; ++j) {
      if (comps.empty() || std::find(comps.begin(), comps.end(), col_comp[j]) == comps.end()) comps.push_back(col_comp[j]);
      std::vector<int>& R = comp_R[col_comp[j]];
      for (int t = bp[j] + col_nbi[j]; t < bp[j + 1]; ++t) R.push_back(brow[t]);
    }
    std::sort(comps.begin(), comps.end());
    for (int c : comps) {
      std::vector<int>& R = comp_R[c];
      std::sort(R.begin(), R.end());
      R.erase(std::unique(R.begin(), R.end()), R.end());
    }
    // update-matrix blocks under construction: one matrix per component (different components of a group may have different parents).
    // Flat records instead of per-block containers (the host plan is rebuilt every tick of the orchestrator: this loop was 75 % of it):
    // a block is found through a dense lower-triangular table over its component's boundary rows, contributions are appended to flat
    // lists and grouped by block with a stable counting sort afterwards -> every list keeps the order it was generated in.
    ub.clear();
    ctab_base.assign(comps.size() + 1, 0);
    for (size_t ci = 0; ci < comps.size(); ++ci) { const int r = (int)comp_R[comps[ci]].size(); ctab_base[ci + 1] = ctab_base[ci] + r * (r + 1) / 2; }
    ctab.assign(ctab_base.back(), -1);
    auto ublock = [&](int comp, int a, int b2) -> int {
      const int ci = (int)(std::lower_bound(comps.begin(), comps.end(), comp) - comps.begin());
      const std::vector<int>& R = comp_R[comp];
      const int ia = (int)(std::lower_bound(R.begin(), R.end(), a) - R.begin()), ib = (int)(std::lower_bound(R.begin(), R.end(), b2) - R.begin());
      int& slot = ctab[ctab_base[ci] + ia * (ia + 1) / 2 + ib];   // a >= b2: both column ids, R ascending
      if (slot < 0) { slot = (int)ub.size(); ub.push_back(UB{comp, a, b2}); }
      return slot;
    };
    own_flat.clear(); src_flat.clear();
    // internal updates (target column in the piece) and own update-matrix contributions (both rows above the piece)
    iu_flat.clear();   // generation order: ascending source column k; rix: per column, position of its rows above the piece in the boundary list of its component
    for (int k = pm.c0; k < pm.c0 + pm.nc; ++k) {
      const int k0 = bp[k] + 1, kin = bp[k] + col_nbi[k], k1 = bp[k + 1];
      const std::vector<int>& Rk = comp_R[col_comp[k]];
      const int tb = ctab_base[std::lower_bound(comps.begin(), comps.end(), col_comp[k]) - comps.begin()];
      rix.resize(k1 - kin);
      for (int q = kin; q < k1; ++q) rix[q - kin] = (int)(std::lower_bound(Rk.begin(), Rk.end(), brow[q]) - Rk.begin());
      for (int pp = k0; pp < k1; ++pp) {
        const int j = brow[pp];
        if (pp < kin) {
          for (int q = pp; q < k1; ++q) {
            const int t = find_blk(j, brow[q]);
            if (t < 0) { out.error = "symbolic factorisation inconsistent (missing fill block)"; return -1; }
            iu_flat.push_back(IU{t, boff[q], boff[pp], k});
          }
        } else {
          for (int q = pp; q < k1; ++q) {
            const int ia = rix[q - kin], ib = rix[pp - kin];   // rows ascend with q: ia >= ib
            int& slot = ctab[tb + ia * (ia + 1) / 2 + ib];
            if (slot < 0) { slot = (int)ub.size(); ub.push_back(UB{col_comp[k], brow[q], j}); }
            own_flat.push_back(OwnRec{slot, boff[q], boff[pp], k});
          }
        }
      }
    }
    // tail pieces: the same internal updates once more, grouped by SOURCE column (right-looking form: a finished column updates every
    // later block of the piece at once -- one tile update deep, where the target-major lists are as deep as the piece has columns)
    pm.pad3 = (int)out.rupd.size();
    if (piece_cls[p] >= 1 && right_ok) {   // tail and mid pieces
      for (int k = pm.c0; k < pm.c0 + pm.nc; ++k) {
        const int k0 = bp[k] + 1, kin = bp[k] + col_nbi[k], k1 = bp[k + 1];
        out.rcol[k].u0 = (int)out.rupd.size() - pm.pad3;
        for (int pp = k0; pp < kin; ++pp) {
          const int j = brow[pp];
          for (int q = pp; q < k1; ++q) {
            const int t = find_blk(j, brow[q]);     // present: checked when the internal updates were listed
            const int tl = out.blk[t].off - pm.lbase, yl = col_yoff[j] - pm.y0;
            if (tl < 0 || tl >= kUpdLocalMax || yl < 0 || yl >= kUpdRightYMax) right_ok = false;   // (an oversized cap_tail) the packed records do not fit: target-major lists everywhere
            const unsigned tpk = (col_dim[brow[t]] == 6 ? kUpdDi6 : 0u) | (t == bp[j] ? kUpdDiag : 0u) | (col_dim[j] == 6 ? kUpdDj6 : 0u) | (col_dim[k] == 6 ? kUpdDk6 : 0u);
            out.rupd.push_back(upd_make_right(boff[q] - pm.lbase, boff[pp] - pm.lbase, tl, yl, tpk));
          }
        }
        out.rcol[k].n = (int)out.rupd.size() - pm.pad3 - out.rcol[k].u0;
      }
    }
    // what the children hand up: absorbed into a column of this piece (assembly) or passed on (update matrix)
    asm_flat.clear();
    for (const URec& u : inbox[p]) {
      if (col_piece[u.b] == p) {
        const int t = find_blk(u.b, u.a);
        if (t < 0) { out.error = "update-matrix block without a target"; return -1; }
        asm_flat.push_back(AsmRec{t, AsmSrc{u.uoff, u.uy}});
      } else {
        if (col_piece[u.a] == p) { out.error = "update-matrix block with its row inside the piece but its column above"; return -1; }
        const int pc = comp_parent[u.comp];   // the component of this group that the sender hangs below
        if (pc < 0 || !std::binary_search(comps.begin(), comps.end(), pc)) { out.error = "update-matrix block routed to the wrong group"; return -1; }
        src_flat.push_back(SrcRec{ublock(pc, u.a, u.b), AsmSrc{u.uoff, u.uy}});
      }
    }
    std::vector<URec>().swap(inbox[p]);
    // group the flat lists (stable): internal updates and assembly sources by target block, own / child contributions by U block
    iu_ptr.assign(pm.nb + 1, 0); as_ptr.assign(pm.nb + 1, 0); own_ptr.assign(ub.size() + 1, 0); src_ptr.assign(ub.size() + 1, 0);
    for (auto& r : iu_flat) iu_ptr[r.t - pm.b0 + 1]++;
    for (auto& r : asm_flat) as_ptr[r.t - pm.b0 + 1]++;
    for (auto& r : own_flat) own_ptr[r.blk + 1]++;
    for (auto& r : src_flat) src_ptr[r.blk + 1]++;
    for (int t = 0; t < pm.nb; ++t) { iu_ptr[t + 1] += iu_ptr[t]; as_ptr[t + 1] += as_ptr[t]; }
    for (size_t q = 0; q < ub.size(); ++q) { own_ptr[q + 1] += own_ptr[q]; src_ptr[q + 1] += src_ptr[q]; }
    iu_s.resize(iu_flat.size()); as_s.resize(asm_flat.size()); src_s.resize(src_flat.size()); own_s.resize(own_flat.size());
    {
      std::vector<int>&c1 = cur1, &c2 = cur2, &c3 = cur3, &c4 = cur4;
      c1.assign(iu_ptr.begin(), iu_ptr.end() - 1); c2.assign(as_ptr.begin(), as_ptr.end() - 1); c3.assign(own_ptr.begin(), own_ptr.end() - 1); c4.assign(src_ptr.begin(), src_ptr.end() - 1);
      for (auto& r : iu_flat) iu_s[c1[r.t - pm.b0]++] = r;
      for (auto& r : asm_flat) as_s[c2[r.t - pm.b0]++] = r.s;
      for (auto& r : own_flat) own_s[c3[r.blk]++] = r;
      for (auto& r : src_flat) src_s[c4[r.blk]++] = r.s;
    }
    // the sources of a target: those of dimension 6 first, then those of dimension 3 (each ascending, as listed) -- the lanes of a wave walk
    // different lists, and a step that meets both kinds pays for both tile updates (tile_update: one branch per kind); with the lists
    // ordered alike the steps of a wave agree on the kind for all but the short 3-wide ends (round 6; the front kernels do the same with
    // their masks).  Any fixed order gives a bitwise repeatable factor.
    for (int t = 0; t < pm.nb; ++t)
      std::stable_partition(iu_s.begin() + iu_ptr[t], iu_s.begin() + iu_ptr[t + 1], [&](const IU& u) { return col_dim[u.k] == 6; });
    for (size_t q = 0; q < ub.size(); ++q)
      std::stable_partition(own_s.begin() + own_ptr[q], own_s.begin() + own_ptr[q + 1], [&](const OwnRec& u) { return col_dim[u.k] == 6; });
    // assembly records, block order
    pm.as0 = (int)out.asrc.size();
    for (int t = pm.b0; t < pm.b0 + pm.nb; ++t) {
      const int n_as = as_ptr[t - pm.b0 + 1] - as_ptr[t - pm.b0];
      if (n_as == 0) continue;
      if (n_as > 255) { out.error = "a block has more than 255 assembly sources"; return -1; }
      out.blk[t].as0 = (int)out.asrc.size() - pm.as0;
      out.blk[t].info |= n_as << kBlkNasShift;
      for (int q = as_ptr[t - pm.b0]; q < as_ptr[t - pm.b0 + 1]; ++q) out.asrc.push_back(as_s[q]);
    }
    pm.nas = (int)out.asrc.size() - pm.as0;
    // internal update records of the piece, level by level / block by block; internal items
    pm.iu0 = (int)out.upd.size();
    bi0.assign(pm.nb, 0); bi1.assign(pm.nb, 0);
    for (int t = pm.b0; t < pm.b0 + pm.nb; ++t) {
      const int j = block_col[t];
      const unsigned tpk = (col_dim[brow[t]] == 6 ? kUpdDi6 : 0u) | (t == bp[j] ? kUpdDiag : 0u) | (col_dim[j] == 6 ? kUpdDj6 : 0u);
      bi0[t - pm.b0] = (int)out.upd.size() - pm.iu0;
      for (int q = iu_ptr[t - pm.b0]; q < iu_ptr[t - pm.b0 + 1]; ++q) {
        const IU& u = iu_s[q];
        out.upd.push_back(upd_make(u.ua - pm.lbase, u.ub2 - pm.lbase, col_yoff[u.k] - pm.y0, tpk | (col_dim[u.k] == 6 ? kUpdDk6 : 0u)));
      }
      bi1[t - pm.b0] = (int)out.upd.size() - pm.iu0;
    }
    pm.nu_i = (int)out.upd.size() - pm.iu0;
    pm.iit0 = (int)out.item.size();
    pm.imb0 = (int)out.mb.size();
    pm.ilv0 = (int)out.ilv.size();
    {
      int c = pm.c0;
      const int cend = pm.c0 + pm.nc;
      while (c < cend) {
        int c1 = c;
        while (c1 < cend && col_il[c1] == col_il[c]) ++c1;
        if (col_il[c] != (int)out.ilv.size() - pm.ilv0) { out.error = "internal levels of a piece are not contiguous"; return -1; }
        ILevel lv{};
        lv.c0 = c; lv.c1 = c1; lv.b0 = bp[c]; lv.b1 = bp[c1];
        lv.it0 = (int)out.item.size() - pm.iit0;
        lv.mb0 = (int)out.mb.size() - pm.imb0;
        // one phase: cut every target's list into items of <= chunk updates; raise the chunk until the partial tiles fit
        int U = 0;
        for (int t = lv.b0; t < lv.b1; ++t) U += bi1[t - pm.b0] - bi0[t - pm.b0];
        const int chunk0 = std::max(opt.min_chunk, (U + slots - 1) / slots);
        lens.clear();
        for (int t = lv.b0; t < lv.b1; ++t) { const int n = bi1[t - pm.b0] - bi0[t - pm.b0]; if (n > chunk0) lens.push_back(n); }
        const int chunk = fit_chunk(chunk0, std::max(chunk0, std::max(U, 1)), pcap);
        int ps = 0;
        for (int t = lv.b0; t < lv.b1; ++t) {
          const int u0 = bi0[t - pm.b0], u1 = bi1[t - pm.b0], n = u1 - u0;
          if (n <= 0) continue;
          const int k = (n + chunk - 1) / chunk;
          const int tloff = out.blk[t].off - pm.lbase;
          const int ylocal = col_yoff[block_col[t]] - pm.y0;
          for (int q = 0; q < k; ++q) {
            const int a2 = u0 + q * chunk, b2 = std::min(u1, a2 + chunk);
            out.item.push_back(ItemMeta{a2, b2 - a2, tloff, (ylocal << kItemYShift) | (k == 1 ? kItemSole : ((ps + q) << kItemSlotShift))});
          }
          if (k > 1) { out.mb.push_back(MbMeta{tloff, ps, k, (out.blk[t].info & 0xFF) | (out.blk[t].info & kBlkDiag) | (ylocal << 12)}); ps += k; }
        }
        piece_pmax[p] = std::max(piece_pmax[p], ps);
        lv.it1 = (int)out.item.size() - pm.iit0;
        lv.mb1 = (int)out.mb.size() - pm.imb0;
        out.ilv.push_back(lv);
        c = c1;
      }
    }
    pm.nilv = (int)out.ilv.size() - pm.ilv0;
    pm.nit_i = (int)out.item.size() - pm.iit0;
    pm.nimb = (int)out.mb.size() - pm.imb0;
    if (pm.nilv > kMaxILevels) { out.error = "a piece has too many internal levels"; return -1; }
    if (pm.ysize >= (1 << (32 - kItemYShift - 1))) { out.error = "a piece has too many unknowns"; return -1; }
    if (pm.lsize >= kUpdLocalMax || pm.ysize >= kUpdYMax) { out.error = "a piece is too large for the packed update records (65535 doubles of L, 8191 unknowns)"; return -1; }
    // update matrix of the piece: blocks in (a, b) order, own update records, child sources, U items
    order.clear();   // (component, a, b) ascending = the order of the dense tables (components and boundary rows ascend)
    for (int slot : ctab) if (slot >= 0) order.push_back(slot);
    pm.uit0 = (int)out.uitem.size();
    pm.umb0 = (int)out.umb.size();
    pm.uu0 = (int)out.upd.size();
    pm.us0 = (int)out.usrc.size();
    {
      int64_t cur = ucur;
      uoffs.assign(ub.size(), 0);
      for (int q : order) { uoffs[q] = (int)cur; cur += blk_doubles(col_dim[ub[q].a], col_dim[ub[q].b]); }
      comp_uy.resize(comps.size());           // Uval offset of a component's rhs part [|R|][6]
      for (size_t ci = 0; ci < comps.size(); ++ci) { comp_uy[ci] = (int)cur; cur += 6 * (int64_t)comp_R[comps[ci]].size(); }
      if (cur >= ((int64_t)1 << 31) - 4096) { out.error = "update matrices too large for int32 offsets"; return -1; }
      const int U = (int)own_s.size();
      const int chunk0 = std::max(opt.min_chunk, (U + slots - 1) / slots);
      lens.clear();
      for (size_t q = 0; q < ub.size(); ++q) { const int n = own_ptr[q + 1] - own_ptr[q]; if (n > chunk0) lens.push_back(n); }
      const int chunk = fit_chunk(chunk0, std::max(chunk0, std::max(U, 1)), pcap);
      int ps = 0;
      const size_t item_first = out.uitem.size();
      for (int c : comps) if (comp_dest[c] >= 0) inbox[comp_dest[c]].reserve(inbox[comp_dest[c]].size() + order.size());   // one growth step per sender
      for (int q : order) {
        const UB& x = ub[q];
        const int di = col_dim[x.a], dj = col_dim[x.b];
        const bool diag = x.a == x.b;
        int uy = -1;
        if (diag) {
          const std::vector<int>& R = comp_R[x.comp];
          uy = comp_uy[std::lower_bound(comps.begin(), comps.end(), x.comp) - comps.begin()] + 6 * (int)(std::lower_bound(R.begin(), R.end(), x.a) - R.begin());
        }
        const unsigned tpk = (di == 6 ? kUpdDi6 : 0u) | (diag ? kUpdDiag : 0u) | (dj == 6 ? kUpdDj6 : 0u);
        const int u0 = (int)out.upd.size() - pm.uu0;
        for (int w = own_ptr[q]; w < own_ptr[q + 1]; ++w) { const OwnRec& u = own_s[w]; out.upd.push_back(upd_make(u.ua - pm.lbase, u.ubo - pm.lbase, col_yoff[u.k] - pm.y0, tpk | (col_dim[u.k] == 6 ? kUpdDk6 : 0u))); }
        const int n = own_ptr[q + 1] - own_ptr[q];
        const int s0 = (int)out.usrc.size() - pm.us0;
        for (int w = src_ptr[q]; w < src_ptr[q + 1]; ++w) out.usrc.push_back(src_s[w]);
        const int ns = src_ptr[q + 1] - src_ptr[q];
        const int k = std::max(1, (n + chunk - 1) / chunk);
        const int shape = (di == 6 ? kUItemDi6 : 0) | (dj == 6 ? kUItemDj6 : 0) | (diag ? kUItemDiag : 0);
        for (int qq = 0; qq < k; ++qq) {
          const int a2 = u0 + qq * chunk, b2 = std::min(u0 + n, a2 + chunk);
          out.uitem.push_back(UItem{a2, std::max(0, b2 - a2), uoffs[q], shape | (k == 1 ? kItemSole : ((ps + qq) << kItemSlotShift)), s0, ns, uy, 0});
        }
        if (k > 1) { out.umb.push_back(UMb{uoffs[q], ps, k, di | (dj << 4) | (diag ? kBlkDiag : 0), s0, ns, uy, 0}); ps += k; }
        if (comp_dest[x.comp] < 0) { out.error = "a root component has an update matrix"; return -1; }
        inbox[comp_dest[x.comp]].push_back(URec{x.a, x.b, uoffs[q], uy, x.comp});
      }
      piece_pmax[p] = std::max(piece_pmax[p], ps);
      ucur = cur;
      // tile-packed order of the items (stable inside a class)
      {
        auto cls = [](const UItem& im) {
          if (!(im.flags & kItemSole)) return 3;
          const int t = ((im.flags & kUItemDi6) ? 2 : 1) * ((im.flags & kUItemDj6) ? 2 : 1);
          return t == 4 ? 0 : (t == 2 ? 1 : 2);
        };
        int cnt[4] = {0, 0, 0, 0};
        for (size_t i = item_first; i < out.uitem.size(); ++i) cnt[cls(out.uitem[i])]++;
        pm.nu4 = cnt[0]; pm.nu2 = cnt[1]; pm.nu1 = cnt[2];
        int at[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
        uit_s.assign(out.uitem.begin() + item_first, out.uitem.end());
        for (const UItem& im : uit_s) out.uitem[item_first + at[cls(im)]++] = im;
      }
    }
    pm.nuit = (int)out.uitem.size() - pm.uit0;
    pm.numb = (int)out.umb.size() - pm.umb0;
    pm.nuu = (int)out.upd.size() - pm.uu0;
    pm.nus = (int)out.usrc.size() - pm.us0;
  }
  out.unz = ucur;
  SSLAM_PT("piece loop")
  if (!right_ok) { out.rupd.clear(); out.rcol.assign(ncol, RCol{0, 0}); }
  // ---- LDS needs (doubles) ---------------------------------------------------------------------------------------------------
  auto lds_f = [&](int p) {
    const PieceMeta& pm = out.piece[p];
    const int ustage = (piece_cls[p] >= 1 || !opt.ustage) ? 0 : 4 * pm.nuit + 4 * pm.numb + ((pm.nuu + 1) & ~1) + pm.nus + 2;
    return 4 * pm.nilv + ((pm.lsize + 1) & ~1) + 2 * ((pm.ysize + 1) & ~1) + 4 * pm.nb + 3 * pm.nc + 2 * pm.nit_i + ((pm.nu_i + 1) & ~1) + 2 * pm.nimb + pm.nas + ustage +
           kItemDoubles * piece_pmax[p] + 8 + (piece_cls[p] >= 1 ? pm.nc + 2 : 0);   // (right-looking form: one RCol per column where the items were)
  };
  auto lds_b = [&](int p) {
    const PieceMeta& pm = out.piece[p];
    return 4 * pm.nilv + 36 * pm.nint + ((pm.ysize + 1) & ~1) + 7 * pm.nb + 3 * pm.nc + 12;
  };
  // ---- launches: one per depth, and a depth with many pieces split by LDS need.  A launch reserves the LDS of its largest piece for
  //      every workgroup, and the pieces of a depth are independent, so they are sorted by need and cut where the number of
  //      workgroups a CU can hold changes (160 KB / 32, 24, 16 workgroups): the typical piece then runs at twice the residency the
  //      largest one of its depth would allow.
  {
    std::vector<int> ptr{0}, launch_order, lnt, lcls;
    launch_order.reserve(out.plv_pieces.size());
    const int cut[3] = {640, 853, 1280};
    for (int l = 0; l < nplv; ++l) {
      // the leaf pieces of the depth (split by LDS need when there are many), then its mid pieces as a launch of their own
      std::vector<int> ps, pmid;
      for (int q = out.plv_ptr[l]; q < out.plv_ptr[l + 1]; ++q) (piece_cls[out.plv_pieces[q]] == 1 ? pmid : ps).push_back(out.plv_pieces[q]);
      const int n = (int)ps.size();
      if (n >= opt.split_min) {
        std::stable_sort(ps.begin(), ps.end(), [&](int a, int b) { return lds_b(a) < lds_b(b); });
        int q0 = 0;
        for (int k = 0; k < 3 && q0 < n; ++k) {
          int q1 = q0;
          while (q1 < n && lds_b(ps[q1]) <= cut[k]) ++q1;
          if (q1 - q0 >= opt.split_min / 2 && n - q1 >= opt.split_min / 2) { for (int q = q0; q < q1; ++q) launch_order.push_back(ps[q]); ptr.push_back((int)launch_order.size()); lnt.push_back(opt.nt_leaf); lcls.push_back(0); q0 = q1; }
        }
        for (int q = q0; q < n; ++q) launch_order.push_back(ps[q]);
      } else {
        for (int p : ps) launch_order.push_back(p);
      }
      if ((int)launch_order.size() > ptr.back()) { ptr.push_back((int)launch_order.size()); lnt.push_back(opt.nt_leaf); lcls.push_back(0); }
      if (!pmid.empty()) { for (int p : pmid) launch_order.push_back(p); ptr.push_back((int)launch_order.size()); lnt.push_back(opt.nt_mid); lcls.push_back(1); }
    }
    out.plv_nt = lnt; out.plv_cls = lcls;
    out.plv_pieces = launch_order;
    out.plv_ptr = ptr;
  }
  const int nlaunch = (int)out.plv_ptr.size() - 1;
  out.lpiece.clear();
  out.lpiece.reserve(npiece);
  for (int p : out.plv_pieces) out.lpiece.push_back(out.piece[p]);
  for (int p : out.tail_pieces) out.lpiece.push_back(out.piece[p]);
  out.plv_lds_f.assign(nlaunch, 0); out.plv_lds_b.assign(nlaunch, 0);
  for (int l = 0; l < nlaunch; ++l)
    for (int q = out.plv_ptr[l]; q < out.plv_ptr[l + 1]; ++q) {
      out.plv_lds_f[l] = std::max(out.plv_lds_f[l], lds_f(out.plv_pieces[q]));
      out.plv_lds_b[l] = std::max(out.plv_lds_b[l], lds_b(out.plv_pieces[q]));
    }
  for (int p : out.tail_pieces) { out.tail_lds_f = std::max(out.tail_lds_f, lds_f(p)); out.tail_lds_b = std::max(out.tail_lds_b, lds_b(p)); }
  // ---- front tables (front_plan.hpp): the same pieces as one blob of relative indices per workgroup -------------------------------
  if (opt.front > 0) {
    FrontHost F;
    front_build(FrontIn{ncol, npiece, ncomp, bp, brow, boff, bsrc, bfmt, col_comp, col_piece, col_dim, col_xoff, col_yoff, col_il, comp_parent, comp_R}, out.piece, out.ilv, F);
    // the kernels keep the diagonal blocks of a level in registers of 8-lane teams, up to two columns per team (front_kernels.hpp)
    for (int p = 0; p < npiece && F.ok; ++p) {
      const int nt = piece_tail[p] ? (opt.nt_tail == 1024 ? 1024 : opt.nt_ftail) : (piece_cls[p] == 1 ? opt.nt_mid : opt.nt_leaf);   // the workgroup k_front_* runs the piece with
      for (int l = 0; l < out.piece[p].nilv; ++l) {
        const ILevel& lv = out.ilv[out.piece[p].ilv0 + l];
        if (lv.c1 - lv.c0 > 2 * (nt / 8)) { F.ok = false; F.why = "a level of a piece has more columns than the workgroup's teams hold"; break; }
      }
    }
    out.front = F.ok; out.front_why = F.why;
    if (F.ok) {
      out.fblob.swap(F.blob); out.fgrp.swap(F.grp); out.funz = F.unz;
      out.lfgrp.reserve(npiece);
      for (int p : out.plv_pieces) out.lfgrp.push_back(out.fgrp[p]);
      for (int p : out.tail_pieces) out.lfgrp.push_back(out.fgrp[p]);
      out.plv_lds_ff.assign(nlaunch, 0);
      for (int l = 0; l < nlaunch; ++l)
        for (int q = out.plv_ptr[l]; q < out.plv_ptr[l + 1]; ++q) out.plv_lds_ff[l] = std::max(out.plv_lds_ff[l], F.lds[out.plv_pieces[q]]);
      for (int p : out.tail_pieces) out.tail_lds_ff = std::max(out.tail_lds_ff, F.lds[p]);
    }
    SSLAM_PT("front tables")
  }
  if (opt.dump) {
    fprintf(stderr, "[chol-dump] B %d cols %d blocks %d lnz %lld unz %lld updates %zu (internal items %zu, U items %zu) column-levels %d pieces %d piece-levels %d tail pieces %zu\n",
            B, ncol, nblk, (long long)lnz, (long long)out.unz, out.upd.size(), out.item.size(), out.uitem.size(), nlev, npiece, nplv, out.tail_pieces.size());
    for (int l = 0; l < nlaunch; ++l)
      fprintf(stderr, "[chol-dump]   launch %d: %d pieces x %d threads, LDS factor %d B backward %d B\n", l, out.plv_ptr[l + 1] - out.plv_ptr[l], out.plv_nt[l],
              out.plv_lds_f[l] * 8, out.plv_lds_b[l] * 8);
    int tmax = 0, tlv = 0;
    for (int g = 0; g < B; ++g) tmax = std::max(tmax, out.tail_ptr[g + 1] - out.tail_ptr[g]);
    for (int q = out.tail_ptr[0]; q < out.tail_ptr[std::min(1, B)]; ++q) tlv += out.piece[out.tail_pieces[q]].nilv;
    fprintf(stderr, "[chol-dump]   tail: <= %d pieces per graph (graph 0: %d internal levels), LDS factor %d B backward %d B\n", tmax, tlv,
            out.tail_lds_f * 8, out.tail_lds_b * 8);
  }
  return 0;
}

}  // namespace sslam
