// HIP kernels of the ps_graph_slam backend for gfx950 (wave64, FP64).
//
// Layout (one device-resident *batch* of B independent graphs, all arrays concatenated by graph):
//   pose[NP][8]   t(3) q(4) pad            lm[NL][4]   point xyz,pad | plane n,d
//   unknown vector x:  [ pose rows: 6 each | landmark rows: 3 each ]   (global over the batch)
//   H values:   Hpp_diag[nPr][36] | Hll_diag[nLr][9] | Hpp_off[nPP][36] | Hpl[nPL][18]   (row-major)
//   edges: structure-of-arrays, SE3 edges and landmark edges separately, grouped by graph
// Reference semantics restated: g2o BlockSolver::buildSystem / computeActiveErrors /
// OptimizationAlgorithmLevenberg behind ps_graph_slam::GraphSLAM::optimize
// (reference src/ps_graph_slam/graph_slam.cpp:182-219).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sslam_math.hpp"

namespace sslam {

constexpr int kEdgeChunk = 256;  // threads per workgroup in edge-parallel kernels
constexpr int kRowChunk = 192;
constexpr int kTileSlots = 96;   // max (row, edge) slots per Jacobian-build tile (LDS budget)   // threads per workgroup in scalar-row kernels (multiple of 6 and 64)

struct GraphSeg {
  int prow0, nprow;  // active pose rows  [prow0, prow0+nprow)
  int lrow0, nlrow;  // active landmark rows
  int eo0, neo;      // SE3 edges
  int el0, nel;      // landmark edges
  int pose0, npose;  // all SE3 vertices (incl. fixed)
  int lm0, nlm;      // all landmark vertices
  int ell0, nell;    // point-point edges (g2o::EdgePointXYZ)
};

struct LmState {
  double lambda, nu, cur_chi, tmp_chi, rho, scale, chi_before, max_diag;
  long long pcg_iters;
  int q, iter, status, in_trial, active, accept, trials, solve_failed;
  int lin, pad0;   // lin: the graph starts a new LM iteration at the next step (its system is rebuilt)
};

// Device view of a batch (plain pointers; passed by value to kernels).
struct BatchView {
  int B, nPr, nLr, nPose, nLm, nEo, nEl, maxRowChunks, maxEdgeChunks;
  const GraphSeg* seg;
  LmState* lm;
  // estimates
  double* pose;        // current
  double* lmk;         // current
  double* pose_trial;
  double* lmk_trial;
  const int* pose_row; // [nPose] -> pose row or -1
  const int* lm_row;   // [nLm]   -> landmark row or -1
  const int* prow_pose;  // [nPr] -> pose index
  const int* lrow_lm;    // [nLr] -> landmark index
  const int* prow_graph; // [nPr]
  const int* lrow_graph; // [nLr]
  const unsigned char* lm_kind;  // [nLm] VT_POINT / VT_PLANE
  // edges
  const int* eo_i; const int* eo_j; const double* eo_z; const double* eo_w; const int* eo_blk;  // eo_blk: off block idx*2 + swap, or -1
  const int* el_p; const int* el_l; const double* el_z; const double* el_w; const int* el_blk;
  // point-point edges (g2o::EdgePointXYZ, reference graph_slam.cpp:168-180): e = (p_b - p_a) - z, Jacobians -I / +I
  int nEll, nLL;                       // edges; unique landmark-landmark blocks
  const int* ell_a; const int* ell_b;  // landmark indices
  const double* ell_z; const double* ell_w;   // SoA: z[3][nEll], upper triangle of Omega [6][nEll]
  const int* ell_id;                   // graph-local edge id (edge-sharded mode)
  const int* llslot_ptr; const int2* llslot_rec;   // per landmark row: {edge, side (0: first vertex, 1: second)}
  const int* llblk_ptr; const int* llblk_edge;     // per landmark-landmark block: its edges
  double* Hll_off;                     // [nLL][9], block (row a < row b) = sum over its edges of -Omega
  // H
  double* Hpp_diag; double* Hll_diag; double* Hpp_off; double* Hpl; double* bvec;
  int64_t h_total;  // doubles in the H allocation
  // row adjacency for SpMV (rows = pose rows then landmark rows)
  const int* adj_ptr; const int* adj_blk; const int* adj_x; const unsigned char* adj_fmt;
  // gather-form Jacobian build: (row, incident edge) slots, pose rows tiled by slot count
  int nTiles;
  const int* tile_row0; const int* tile_row1;       // [nTiles] pose-row range of a tile (one graph each)
  const int* pslot_ptr;                             // [nPr+1]
  const int4* pslot_rec;                            // [slots] {edge, kind, pose i | pose, pose j | landmark}
  const int* pslot_edge; const unsigned char* pslot_kind;  // kind 0: SE3 edge, self = i; 1: SE3, self = j; 2: landmark edge
  const int* lslot_ptr; const int* lslot_edge;      // [nLr+1], landmark-side slots (edge index into el_*)
  int nDupEo, nDupEl; const int* dup_eo; const int* dup_el;  // non-owner edges of shared off-diagonal blocks
  // edge-sharded mode (one graph's edges split across ranks, SURVEY 8e mode E): a rank builds the partial normal equations of the
  // edges whose graph-local id lies in [shard_lo[g], shard_hi[g]); the partial [H || b] arrays are summed with one all-reduce
  const int* eo_id; const int* el_id;        // graph-local edge id of every SE3 / landmark edge
  const int* shard_lo; const int* shard_hi;  // [B]
  int dbg = 0;                               // SSLAM_LIN_DBG: timing experiments only (results are wrong when set)
  double dcs_phi = 0.0;                      // > 0: RobustKernelDCS on the landmark edges (chi2 uses rho[0], Omega is scaled by rho[1])
  // PCG vectors
  double* x; double* r; double* z; double* p; double* q; double* Minv;  // Minv: [nPr*36 | nLr*9]
  double* part_a; double* part_b; double* part_c;  // [B*maxChunks] partial sums
  double* rz;       // [2][B]
  double* bb;       // [B]
  int* pcg_done;    // [2][B] double-buffered by iteration parity
  int* pcg_fail;    // [B]
  int* flags;       // [0] any_in_trial, [1] all_pcg_done
};

// g2o::RobustKernelDCS::robustify: rho[1], the factor on Omega; rho[0] = rho[1] * e2
__device__ __forceinline__ double dcs_rho1(double phi, double e2) {
  const double scale = (2.0 * phi) / (phi + e2);
  return scale >= 1.0 ? 1.0 : scale * scale;
}
__device__ __forceinline__ double quad3(const double W[9], const double e[3]) {
  double c = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) c += e[r] * (W[r * 3 + 0] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
  return c;
}

// ------------------------------------------------------------------------------------------
// deterministic workgroup reduction: wave shuffles, then a fixed-order sum of the wave partials
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* lds /* NT/64 doubles */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int k = 0; k < NT / 64; ++k) s += lds[k];
  __syncthreads();
  return s;  // valid on thread 0
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int k = 0; k < NT / 64; ++k) s = fmax(s, lds[k]);
  __syncthreads();
  return s;
}
// fixed-order sum of n partials by one wave (all lanes get the result)
__device__ __forceinline__ double wave_sum_partials(const double* part, int n) {
  const int lane = threadIdx.x & 63;
  double s = 0;
  for (int k = lane; k < n; k += 64) s += part[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

__device__ __forceinline__ Pose load_pose(const double* a, int idx) {
  const double* p = a + (size_t)idx * 8;
  return Pose{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}};
}
__device__ __forceinline__ void store_pose(double* a, int idx, const Pose& P) {
  double* p = a + (size_t)idx * 8;
  p[0] = P.t.x; p[1] = P.t.y; p[2] = P.t.z; p[3] = P.q.x; p[4] = P.q.y; p[5] = P.q.z; p[6] = P.q.w;
}
// symmetric 6x6 from 21 upper-triangular entries stored SoA with stride n: w[k*n + e]
__device__ __forceinline__ void load_sym6(const double* w, int n, int e, double W[36]) {
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) { const double v = w[(size_t)k * n + e]; W[r * 6 + c] = v; W[c * 6 + r] = v; ++k; }
}
__device__ __forceinline__ void load_sym3(const double* w, int n, int e, double W[9]) {
  int k = 0;
  for (int r = 0; r < 3; ++r)
    for (int c = r; c < 3; ++c) { const double v = w[(size_t)k * n + e]; W[r * 3 + c] = v; W[c * 3 + r] = v; ++k; }
}

}  // namespace sslam
