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
  const int* prow_perm;  // [nPr] thread -> pose row of k_linearize_rowthread: the rows of a graph ordered by their slot counts (EdgeSE3 slots, then
                         //       landmark slots), so that the lanes of a wave walk slot lists of the same shape (round 6)
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
  // plane landmarks (round 6): error and central-difference Jacobians of every EdgeSE3Plane, evaluated ONCE per linearisation by
  // k_plane_jacobians (a thread per (edge, perturbation)) -- [nEl][30] = {e 3 | J_l 3x3 | J_i 3x6}; nullptr when the batch has no plane
  double* pj;
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

// ---- scalar-row helpers -------------------------------------------------------------------------
struct RowRef {
  int valid;  // inside the graph's range
  int is_pose;
  int row;    // pose row or landmark row (global)
  int r;      // component inside the block
  int xoff;   // offset in the unknown vector
  int base;   // xoff of component 0
};
__device__ __forceinline__ RowRef row_ref(const BatchView& V, const GraphSeg& sg, int e) {
  RowRef R;
  const int npd = sg.nprow * 6;
  R.valid = e < npd + sg.nlrow * 3;
  if (e < npd) {
    R.is_pose = 1; R.row = sg.prow0 + e / 6; R.r = e % 6; R.base = 6 * R.row;
  } else {
    const int u = e - npd;
    R.is_pose = 0; R.row = sg.lrow0 + u / 3; R.r = u % 3; R.base = 6 * V.nPr + 3 * R.row;
  }
  R.xoff = R.base + R.r;
  return R;
}

__device__ __forceinline__ int row_chunks(const GraphSeg& sg) { return (sg.nprow * 6 + sg.nlrow * 3 + kRowChunk - 1) / kRowChunk; }
__device__ __forceinline__ int edge_chunks(const GraphSeg& sg) { return (sg.neo + sg.nel + sg.nell + kEdgeChunk - 1) / kEdgeChunk; }


// x [+] dx of one block row t (pose rows first, then landmark rows) of a graph that is in a trial -> the trial estimates
// (VertexSE3 / VertexPointXYZ / VertexPlane::oplus)
__device__ __forceinline__ void oplus_row(const BatchView& V, int t, const double* __restrict__ dx) {
  if (t < V.nPr) {
    const int g = V.prow_graph[t];
    if (!V.lm[g].in_trial) return;
    const int pi = V.prow_pose[t];
    double d[6];
    for (int k = 0; k < 6; ++k) d[k] = dx[6 * (size_t)t + k];
    store_pose(V.pose_trial, pi, se3_oplus(load_pose(V.pose, pi), d));
  } else if (t < V.nPr + V.nLr) {
    const int l = t - V.nPr;
    const int g = V.lrow_graph[l];
    if (!V.lm[g].in_trial) return;
    const int li = V.lrow_lm[l];
    const double* d = dx + 6 * (size_t)V.nPr + 3 * (size_t)l;
    const double* c = V.lmk + (size_t)li * 4;
    double* o = V.lmk_trial + (size_t)li * 4;
    if (V.lm_kind[li] == VT_POINT) {
      o[0] = c[0] + d[0]; o[1] = c[1] + d[1]; o[2] = c[2] + d[2]; o[3] = 0;
    } else {
      const double dv[3] = {d[0], d[1], d[2]};
      const Plane P = pl_oplus(Plane{{c[0], c[1], c[2]}, c[3]}, dv);
      o[0] = P.n.x; o[1] = P.n.y; o[2] = P.n.z; o[3] = P.d;
    }
  }
}

// an accepted trial becomes the estimate: block row t of its graph
__device__ __forceinline__ void commit_row(const BatchView& V, int t) {
  if (t < V.nPr) {
    const int g = V.prow_graph[t];
    if (!V.lm[g].accept) return;
    const int pi = V.prow_pose[t];
    for (int k = 0; k < 7; ++k) V.pose[(size_t)pi * 8 + k] = V.pose_trial[(size_t)pi * 8 + k];
  } else if (t < V.nPr + V.nLr) {
    const int l = t - V.nPr;
    const int g = V.lrow_graph[l];
    if (!V.lm[g].accept) return;
    const int li = V.lrow_lm[l];
    for (int k = 0; k < 4; ++k) V.lmk[(size_t)li * 4 + k] = V.lmk_trial[(size_t)li * 4 + k];
  }
}

// g2o OptimizationAlgorithmLevenberg: accept / reject one damping trial of a graph; one thread
__device__ __forceinline__ void lm_control_apply(LmState& S, double tchi, double sc, int solve_failed, int max_iters) {
  double tmp = tchi;
  double scale = sc + 1e-3;
  if (solve_failed) { tmp = INFINITY; scale = 1.0; S.solve_failed += 1; }
  const double rho = (S.cur_chi - tmp) / scale;
  S.tmp_chi = tmp; S.scale = scale; S.rho = rho; S.trials += 1;
  if (rho > 0 && isfinite(tmp)) {
    double a = 2 * rho - 1;
    double alpha = 1.0 - a * a * a;
    alpha = fmin(alpha, 2.0 / 3.0);
    const double sf = fmax(1.0 / 3.0, alpha);
    S.lambda *= sf; S.nu = 2; S.cur_chi = tmp; S.accept = 1;
  } else {
    S.lambda *= S.nu; S.nu *= 2; S.accept = 0;
  }
  S.q += 1;
  const int again = (rho < 0 && S.q < 10);
  S.in_trial = again;   // a rejected trial is repeated at the next step with the raised lambda
  if (!again) {
    S.iter += 1;
    if (S.q == 10 || rho == 0) { S.status = 1; S.active = 0; }
    else if (S.iter >= max_iters) { S.status = 0; S.active = 0; }
    else S.lin = 1;     // next step: new linearisation, new iteration
  }
}

// chi2 term of the graph-local edge e of graph segment sg (SE3 edges, then landmark edges, then point-point edges; 0 beyond the last):
// g2o computeActiveErrors + the robust kernel's rho[0]
__device__ __forceinline__ double edge_chi2(const BatchView& V, const GraphSeg& sg, int e, const double* __restrict__ pose, const double* __restrict__ lmk) {
  double c = 0;
  if (e < sg.neo) {
    const int k = sg.eo0 + e;
    const Pose Xi = load_pose(pose, V.eo_i[k]), Xj = load_pose(pose, V.eo_j[k]);
    const int n = V.nEo;
    const Pose Z{{V.eo_z[0 * (size_t)n + k], V.eo_z[1 * (size_t)n + k], V.eo_z[2 * (size_t)n + k]},
                 {V.eo_z[3 * (size_t)n + k], V.eo_z[4 * (size_t)n + k], V.eo_z[5 * (size_t)n + k], V.eo_z[6 * (size_t)n + k]}};
    Se3Lin L;
    se3_error(Xi, Xj, Z, L);
    double W[36];
    load_sym6(V.eo_w, n, k, W);
    for (int r = 0; r < 6; ++r) {
      double a = 0;
      for (int s = 0; s < 6; ++s) a += W[r * 6 + s] * L.e[s];
      c += L.e[r] * a;
    }
  } else if (e < sg.neo + sg.nel) {
    const int k = sg.el0 + (e - sg.neo);
    const int n = V.nEl;
    const int li = V.el_l[k];
    const Pose Xi = load_pose(pose, V.el_p[k]);
    const double* lp = lmk + (size_t)li * 4;
    double err[3];
    if (V.lm_kind[li] == VT_POINT) {
      PointLin L;
      point_error(Xi, Vec3{lp[0], lp[1], lp[2]},
                  Vec3{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, L);
      err[0] = L.e[0]; err[1] = L.e[1]; err[2] = L.e[2];
    } else {
      plane_error(Xi, Plane{{lp[0], lp[1], lp[2]}, lp[3]},
                  Plane{{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, V.el_z[3 * (size_t)n + k]}, err);
    }
    double W[9];
    load_sym3(V.el_w, n, k, W);
    for (int r = 0; r < 3; ++r) {
      double a = 0;
      for (int s = 0; s < 3; ++s) a += W[r * 3 + s] * err[s];
      c += err[r] * a;
    }
    if (V.dcs_phi > 0) c *= dcs_rho1(V.dcs_phi, c);
  } else if (e < sg.neo + sg.nel + sg.nell) {   // g2o::EdgePointXYZ: e = (p_b - p_a) - z
    const int k = sg.ell0 + (e - sg.neo - sg.nel);
    const size_t n = V.nEll;
    const double* pa = lmk + (size_t)V.ell_a[k] * 4;
    const double* pb = lmk + (size_t)V.ell_b[k] * 4;
    double err[3], W[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) err[r] = (pb[r] - pa[r]) - V.ell_z[r * n + k];
    load_sym3(V.ell_w, (int)n, k, W);
    c = quad3(W, err);
  }
  return c;
}

}  // namespace sslam
