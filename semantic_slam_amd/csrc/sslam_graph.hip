// MI355X-native backend of ps_graph_slam::GraphSLAM — kernels, batch engine and C-ABI.
// See include/sslam.h for the boundary and DESIGN.md for the data layout / roofline accounting.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is bound at run time (rccl_api below)
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>

#include "../../include/sslam.h"
#include "graph_host.hpp"
#include "graph_kernels.hpp"
#include "sslam_common.hpp"
#include "graph_engine.hpp"
#include "chol_plan.hpp"

namespace sslam {

// =============================================================================================
// Kernels
// =============================================================================================

// ---- residual / chi2 (g2o computeActiveErrors): per-graph partial sums, deterministic ---------
__global__ __launch_bounds__(kEdgeChunk) void k_chi2(BatchView V, const double* __restrict__ pose,
                                                    const double* __restrict__ lmk, int mask_mode,
                                                    double* __restrict__ part) {
  __shared__ double red[kEdgeChunk / 64];
  const int g = blockIdx.y;
  const LmState& S = V.lm[g];
  if (mask_mode == 1 && !S.in_trial) return;
  if (mask_mode == 2 && !S.active) return;
  const GraphSeg sg = V.seg[g];
  const int e = blockIdx.x * kEdgeChunk + threadIdx.x;
  if (blockIdx.x * kEdgeChunk >= sg.neo + sg.nel + sg.nell) return;
  const double c = edge_chi2(V, sg, e, pose, lmk);
  const double s = block_sum<kEdgeChunk>(c, red);
  if (threadIdx.x == 0) part[(size_t)g * V.maxEdgeChunks + blockIdx.x] = s;
}

// off-diagonal block codes: >= 0 owner edge, <= -2 a further edge on the same vertex pair, -1 none
__device__ __forceinline__ int decode_blk(int code) { return code <= -2 ? -2 - code : code; }
__device__ __forceinline__ Pose load_meas_pose(const double* z, int n, int k) {
  return Pose{{z[0 * (size_t)n + k], z[1 * (size_t)n + k], z[2 * (size_t)n + k]},
              {z[3 * (size_t)n + k], z[4 * (size_t)n + k], z[5 * (size_t)n + k], z[6 * (size_t)n + k]}};
}
struct alignas(16) D2 { double a, b; };

// ---- Jacobian build (gather form, deterministic, no atomics): one THREAD per pose row ---------------
// No redundancy across lanes: a thread evaluates each incident edge once, keeps the 6x6 Jacobian of
// its own vertex in registers, streams the other vertex's columns for the off-diagonal block it
// owns, and accumulates the symmetric diagonal block (21 values) + b (6) in a thread-private LDS
// column (lane-contiguous layout -> conflict free), written to HBM once at the end.  All block stores
// are 16-byte (two doubles): half the write requests of scalar stores at a 288-byte lane stride.
constexpr int kRowThreads = 64;
__device__ __forceinline__ int tri21(int a, int c) { return a * 6 - (a * (a - 1)) / 2 + (c - a); }   // a <= c
__device__ __forceinline__ void store2(double* p, double a, double b) { *reinterpret_cast<D2*>(p) = D2{a, b}; }

__device__ __forceinline__ Pose load_pose16(const double* a, int idx) {   // four 16-byte loads of the 64-byte record
  const D2* p = reinterpret_cast<const D2*>(a + (size_t)idx * 8);
  const D2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
  return Pose{{v0.a, v0.b, v1.a}, {v1.b, v2.a, v2.b, v3.a}};
}

// EdgeSE3Plane (reference include/g2o/edge_se3_plane.hpp:8-48; g2o's BaseBinaryEdge numeric Jacobian: central differences, delta 1e-9): 19
// evaluations of the error per edge -- the error itself, +/- delta on the six components of the pose and on the three of the plane -- each ~900
// FP64 instructions (four atan2, two sincos pairs).  Until round 6 the pose row's thread did all 19 for every plane slot (at one wave per SIMD:
// the kernel holds 304 VGPRs) and the landmark row's lanes the 7 of the plane side again: 15.6 ms per 512-graph build against 2.1 ms with
// point landmarks.  Now a thread per (edge, evaluation) -- a pair of + / - evaluations sits in neighbouring lanes and meets through one shuffle --,
// full occupancy, every evaluation done once; the row kernels read the 30 doubles.  Same functions on the same inputs: the values the row kernels used to compute themselves.
constexpr int kPjDoubles = 30;   // {e 3 | J_l 3 x 3 row-major | J_i 3 x 6 row-major}
// Two regions of workgroups so that a wave runs ONE of the two perturbation paths (se3_oplus / pl_oplus: ~150 / ~300 instructions next to the
// ~400 of the error itself; mixed in one wave both were paid by every lane: 4.84 -> 3.5 ms per 512 graphs): blocks [0, blocksA) hold fourteen
// lanes per edge -- lane 0 the error, lane 1 idle, lanes 2 + 2 d / 3 + 2 d the + / - evaluations of pose component d --, the blocks behind them
// six lanes per edge, the + / - evaluations of the three plane components.  A pair sits in neighbouring lanes (even lane counts per edge).
__global__ __launch_bounds__(256) void k_plane_jacobians(BatchView V, int blocksA) {
  const bool regA = (int)blockIdx.x < blocksA;
  const int per = regA ? 14 : 6;
  const long long t = (long long)(regA ? blockIdx.x : blockIdx.x - blocksA) * 256 + threadIdx.x;
  const long long e64 = t / per;
  const int j = (int)(t - e64 * per);
  const bool in = e64 < V.nEl;
  const int e = in ? (int)e64 : 0;
  const int pi = V.el_p[e], li = V.el_l[e];
  bool live = in && V.lm_kind[li] == VT_PLANE && !(regA && j == 1);
  if (live) {
    const int pr = V.pose_row[pi], lr = V.lm_row[li];
    const int g = pr >= 0 ? V.prow_graph[pr] : (lr >= 0 ? V.lrow_graph[lr] : -1);
    live = g >= 0 && V.lm[g].lin;
  }
  const double delta = 1e-9;
  double err[3] = {0, 0, 0};
  if (live) {
    const size_t n = (size_t)V.nEl;
    Pose Xi = load_pose(V.pose, pi);
    const double* lp = V.lmk + (size_t)li * 4;
    Plane pw{{lp[0], lp[1], lp[2]}, lp[3]};
    const Plane z{{V.el_z[0 * n + e], V.el_z[1 * n + e], V.el_z[2 * n + e]}, V.el_z[3 * n + e]};
    const double sd = (j & 1) ? -delta : delta;
    if (regA) {
      if (j >= 2) {
        const int d = (j - 2) >> 1;
        double dv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dv[q] = q == d ? sd : 0.0;
        Xi = se3_oplus(Xi, dv);
      }
    } else {
      const int d = j >> 1;
      double d3[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) d3[q] = q == d ? sd : 0.0;
      pw = pl_oplus(pw, d3);
    }
    plane_error(Xi, pw, z, err);
  }
  // the - evaluation of a pair sits one lane up
  const double m0 = __shfl_down(err[0], 1, 64), m1 = __shfl_down(err[1], 1, 64), m2 = __shfl_down(err[2], 1, 64);
  if (!live || (j & 1)) return;
  double* o = V.pj + (size_t)e * kPjDoubles;
  if (regA && j == 0) { o[0] = err[0]; o[1] = err[1]; o[2] = err[2]; return; }
  const double scalar = 1.0 / (2 * delta);
  const double c0 = scalar * (err[0] - m0), c1 = scalar * (err[1] - m1), c2 = scalar * (err[2] - m2);
  if (regA) { const int d = (j - 2) >> 1; o[12 + d] = c0; o[18 + d] = c1; o[24 + d] = c2; }
  else { const int d = j >> 1; o[3 + d] = c0; o[6 + d] = c1; o[9 + d] = c2; }
}

// Pose-row kernel: every slot (incident edge) of a row evaluated by the row's own thread, contributions summed in slot order.
// What rounds 2-4 built and measured against it -- all correct, none faster, all removed from the library in round 5 (DESIGN.md section 5
// keeps the numbers): a hand-over form that evaluates a chain edge once (1.97 vs 2.00 ms per 512-graph build, but it adds the handed block
// last and LM's accept / reject decisions at convergence follow the last bit of H: 13 extra rejected trials per 20 iterations); two
// role-specialised waves per 64-row tile (216 VGPRs, two waves per SIMD: 2.23 vs 1.89 ms); EdgeSE3 and landmark slots in two launches
// (2.52 vs 2.01 ms); a row-wise EdgeSE3 form with 222 VGPRs (2.04 vs 1.96 ms); the kernel compiled for two / three waves per SIMD (50 / 164
// spilled registers: 2.46 / 4.26 vs 2.12 ms); staged, coalesced block stores, also with the next round's operands requested before the
// stores (2.06 vs 2.02, 2.09 vs 2.07 ms).  What they established: the build is bound by the life time of ONE wave per SIMD (304 VGPRs: both
// Jacobians 45, Omega 27, J^T Omega 36, an off-diagonal block 36 doubles) -- instruction issue of a single FP64 wave plus the round trips it
// cannot overlap -- not by bytes, occupancy alone, or the shape of its stores.  Round 5 closed the occupancy question: the SAME arithmetic
// with shorter live ranges (J_j built where the off-diagonal block is formed, that block two rows at a time, the own pose re-fetched per
// slot) compiles to 231 VGPRs -- two waves per SIMD, no spill -- and ran 2.33 vs 2.10 ms per 512-graph build.  By the raw counters the build
// moves ~13 MB per graph (both endpoint rows read an EdgeSE3, the landmark rows read every landmark edge again) at ~3.2 TB/s of mixed reads
// and scattered 16-byte writes: it sits at what HBM gives such a mix, and only fewer bytes would make it faster.
template <bool PL, bool SHARD>
__global__ __launch_bounds__(kRowThreads, 1) void k_linearize_rowthread(BatchView V) {
  __shared__ double accD[27][kRowThreads];   // this row's diagonal block (upper triangle) and rhs: a thread-private LDS column, conflict free
  const int tid = threadIdx.x;
  const int slot_t = blockIdx.x * kRowThreads + tid;
  if (slot_t >= V.nPr) return;
  const int row = V.prow_perm[slot_t];   // (rows of equal slot counts share a wave)
  if (!V.lm[V.prow_graph[row]].lin) return;
#pragma unroll
  for (int k = 0; k < 27; ++k) accD[k][tid] = 0.0;
  const int s0 = V.pslot_ptr[row], s1 = V.pslot_ptr[row + 1];
  const int own = V.prow_pose[row];
  const int sh_lo = SHARD ? V.shard_lo[V.prow_graph[row]] : 0, sh_hi = SHARD ? V.shard_hi[V.prow_graph[row]] : 0;
  const Pose Xown = load_pose16(V.pose, own);   // this row's vertex: loaded once, reused by every slot
  for (int s = s0; s < s1; ++s) {
    const int4 rec = V.pslot_rec[s];
    const int e = rec.x, kind = rec.y & 15, ia = rec.z, ib = rec.w;
    if (kind != 2) {
      // EdgeSE3: both Jacobians are 2 x 2 block upper triangular in 3 x 3 blocks,
      //   J_i = [[-Ra, 2 Ra [tb]x], [0, Ci]],   J_j = [[Re, 0], [0, Fj]],
      // so J^T Omega J is formed from 3 x 3 products of the blocks (half the FMAs and a smaller live set than dense 6 x 6).
      const int n = V.nEo;
      const bool iside = (kind == 0);
      Se3Lin L;
      const Pose Xo = load_pose16(V.pose, iside ? ib : ia);
      const Pose Zm = load_meas_pose(V.eo_z, n, e);
      se3_error(iside ? Xown : Xo, iside ? Xo : Xown, Zm, L);
      double P[9], Q[9], R[9];   // Omega = [[P, Q], [Q^T, R]]
      // edge-sharded mode: an edge outside this rank's range contributes nothing (everything below is linear in Omega; the owner
      // of an off-diagonal block still writes it, as zeros)
      const int eid = SHARD ? V.eo_id[e] : 0;
      const double mk = (!SHARD || (eid >= sh_lo && eid < sh_hi)) ? 1.0 : 0.0;
      auto w21 = [&](int q) { return V.eo_w[(size_t)q * n + e]; };
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          P[r * 3 + c] = mk * w21(r <= c ? tri21(r, c) : tri21(c, r));
          Q[r * 3 + c] = mk * w21(tri21(r, 3 + c));
          R[r * 3 + c] = mk * w21(r <= c ? tri21(3 + r, 3 + c) : tri21(3 + c, 3 + r));
        }
      double A[9], B[9], Cc[9];   // own Jacobian [[A, B], [0, Cc]]  (B = 0 on the j side)
      if (iside) {
        const Vec3 tb = L.tb;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double r0 = L.Ra.m[r * 3], r1 = L.Ra.m[r * 3 + 1], r2 = L.Ra.m[r * 3 + 2];
          A[r * 3] = -r0; A[r * 3 + 1] = -r1; A[r * 3 + 2] = -r2;
          B[r * 3 + 0] = 2 * (r1 * tb.z - r2 * tb.y);     // Ra * (0, tz, -ty)
          B[r * 3 + 1] = 2 * (-r0 * tb.z + r2 * tb.x);    // Ra * (-tz, 0, tx)
          B[r * 3 + 2] = 2 * (r0 * tb.y - r1 * tb.x);     // Ra * (ty, -tx, 0)
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const Quat vk = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0, 0.0};
          const Quat t = qmul(qmul(L.qa, vk), L.qb);
          Cc[0 * 3 + k] = -L.s * t.x; Cc[1 * 3 + k] = -L.s * t.y; Cc[2 * 3 + k] = -L.s * t.z;
        }
      }
      double E[9], F[9];          // J_j = [[E, 0], [0, F]]
      {
        const Mat3 Re = qmat(L.qe);
        const double w = L.s * L.qe.w, x = L.s * L.qe.x, y = L.s * L.qe.y, z = L.s * L.qe.z;
#pragma unroll
        for (int q = 0; q < 9; ++q) E[q] = Re.m[q];
        F[0] = w; F[1] = -z; F[2] = y; F[3] = z; F[4] = w; F[5] = -x; F[6] = -y; F[7] = x; F[8] = w;
      }
      if (!iside) {
#pragma unroll
        for (int q = 0; q < 9; ++q) { A[q] = E[q]; B[q] = 0.0; Cc[q] = F[q]; }
      }
      // M = J_self^T Omega = [[A^T P, A^T Q], [B^T P + C^T Q^T, B^T Q + C^T R]]
      double M11[9], M12[9], M21[9], M22[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double m11 = 0, m12 = 0, m21 = 0, m22 = 0;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            m11 += A[r * 3 + a] * P[r * 3 + c];
            m12 += A[r * 3 + a] * Q[r * 3 + c];
            m21 += Cc[r * 3 + a] * Q[c * 3 + r];
            m22 += Cc[r * 3 + a] * R[r * 3 + c];
          }
          if (iside) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { m21 += B[r * 3 + a] * P[r * 3 + c]; m22 += B[r * 3 + a] * Q[r * 3 + c]; }
          }
          M11[a * 3 + c] = m11; M12[a * 3 + c] = m12; M21[a * 3 + c] = m21; M22[a * 3 + c] = m22;
        }
      const int blk = iside ? V.eo_blk[e] : -1;
      if (blk >= 0) {   // owner of the off-diagonal block: J_i^T Omega J_j = [[M11 E, M12 F], [M21 E, M22 F]]
        double* O = V.Hpp_off + (size_t)(blk >> 1) * 36;
        const bool swapped = blk & 1;
        double o[36];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            double o11 = 0, o12 = 0, o21 = 0, o22 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              o11 += M11[a * 3 + r] * E[r * 3 + c]; o12 += M12[a * 3 + r] * F[r * 3 + c];
              o21 += M21[a * 3 + r] * E[r * 3 + c]; o22 += M22[a * 3 + r] * F[r * 3 + c];
            }
            o[a * 6 + c] = o11; o[a * 6 + 3 + c] = o12; o[(3 + a) * 6 + c] = o21; o[(3 + a) * 6 + 3 + c] = o22;
          }
        if (!swapped) {   // stored [row_i][row_j]
#pragma unroll
          for (int k = 0; k < 36; k += 2) store2(O + k, o[k], o[k + 1]);
        } else {          // stored transposed
#pragma unroll
          for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int a = 0; a < 6; a += 2) store2(O + c * 6 + a, o[a * 6 + c], o[(a + 1) * 6 + c]);
        }
      }
      // diagonal block J^T Omega J (upper triangle) and b -= J^T Omega e
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double d11 = 0, d12 = 0, d22 = 0;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            d11 += M11[a * 3 + r] * A[r * 3 + c];
            d12 += M12[a * 3 + r] * Cc[r * 3 + c];
            d22 += M22[a * 3 + r] * Cc[r * 3 + c];
          }
          if (iside) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { d12 += M11[a * 3 + r] * B[r * 3 + c]; d22 += M21[a * 3 + r] * B[r * 3 + c]; }
          }
          if (a <= c) { accD[tri21(a, c)][tid] += d11; accD[tri21(3 + a, 3 + c)][tid] += d22; }
          accD[tri21(a, 3 + c)][tid] += d12;
        }
        accD[21 + a][tid] -= M11[a * 3] * L.e[0] + M11[a * 3 + 1] * L.e[1] + M11[a * 3 + 2] * L.e[2] +
                             M12[a * 3] * L.e[3] + M12[a * 3 + 1] * L.e[4] + M12[a * 3 + 2] * L.e[5];
        accD[24 + a][tid] -= M21[a * 3] * L.e[0] + M21[a * 3 + 1] * L.e[1] + M21[a * 3 + 2] * L.e[2] +
                             M22[a * 3] * L.e[3] + M22[a * 3 + 1] * L.e[4] + M22[a * 3 + 2] * L.e[5];
      }
    } else {
      const int n = V.nEl;
      const Pose Xi = Xown;
      const double* lp = V.lmk + (size_t)ib * 4;
      auto zl = [&](int k) { return V.el_z[k * (size_t)n + e]; };
      double err[3], Ji[18], Jl[9];   // Ji row-major 3x6, Jl row-major 3x3
      if (!PL || V.lm_kind[ib] == VT_POINT) {
        PointLin L;
        point_error(Xi, Vec3{lp[0], lp[1], lp[2]}, Vec3{zl(0), zl(1), zl(2)}, L);
        point_jacobians(L, Ji, Jl);
        err[0] = L.e[0]; err[1] = L.e[1]; err[2] = L.e[2];
      } else {
        // error and both Jacobians from k_plane_jacobians (fifteen 16-byte loads instead of nineteen evaluations)
        const D2* pj = reinterpret_cast<const D2*>(V.pj + (size_t)e * kPjDoubles);
        double v[kPjDoubles];
#pragma unroll
        for (int q = 0; q < kPjDoubles / 2; ++q) { const D2 w = pj[q]; v[2 * q] = w.a; v[2 * q + 1] = w.b; }
#pragma unroll
        for (int q = 0; q < 3; ++q) err[q] = v[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) Jl[q] = v[3 + q];
#pragma unroll
        for (int q = 0; q < 18; ++q) Ji[q] = v[12 + q];
      }
      double W[9];
      load_sym3(V.el_w, n, e, W);
      double dcs = 1.0;
      if (V.dcs_phi > 0) dcs = dcs_rho1(V.dcs_phi, quad3(W, err));   // from the unmasked Omega: every rank scales its share alike
      const int lid = SHARD ? V.el_id[e] : 0;
      if (SHARD && !(lid >= sh_lo && lid < sh_hi)) {
#pragma unroll
        for (int q = 0; q < 9; ++q) W[q] = 0.0;
      }
      if (V.dcs_phi > 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) W[q] *= dcs;
      }
      double WJi[18], We[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        We[a] = W[a * 3 + 0] * err[0] + W[a * 3 + 1] * err[1] + W[a * 3 + 2] * err[2];
#pragma unroll
        for (int c = 0; c < 6; ++c) WJi[a * 6 + c] = W[a * 3 + 0] * Ji[c] + W[a * 3 + 1] * Ji[6 + c] + W[a * 3 + 2] * Ji[12 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int a = 0; a <= c; ++a) accD[tri21(a, c)][tid] += Ji[a] * WJi[c] + Ji[6 + a] * WJi[6 + c] + Ji[12 + a] * WJi[12 + c];
        accD[21 + c][tid] -= Ji[c] * We[0] + Ji[6 + c] * We[1] + Ji[12 + c] * We[2];
      }
      const int blk = V.el_blk[e];
      if (blk >= 0) {
        double WJl[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) WJl[a * 3 + c] = W[a * 3 + 0] * Jl[c] + W[a * 3 + 1] * Jl[3 + c] + W[a * 3 + 2] * Jl[6 + c];
        double o[18];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) o[a * 3 + c] = Ji[a] * WJl[c] + Ji[6 + a] * WJl[3 + c] + Ji[12 + a] * WJl[6 + c];
        double* O = V.Hpl + (size_t)blk * 18;
#pragma unroll
        for (int k = 0; k < 18; k += 2) store2(O + k, o[k], o[k + 1]);
      }
    }
  }
  double* P = V.Hpp_diag + (size_t)row * 36;
  double* Bv = V.bvec + (size_t)row * 6;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; c += 2)
      store2(P + a * 6 + c, accD[a <= c ? tri21(a, c) : tri21(c, a)][tid], accD[a <= c + 1 ? tri21(a, c + 1) : tri21(c + 1, a)][tid]);
#pragma unroll
  for (int c = 0; c < 6; c += 2) store2(Bv + c, accD[21 + c][tid], accD[22 + c][tid]);
}

// landmark rows: 16 lanes per landmark, each lane walks a strided subset of the incident edges,
// butterfly reduction in a fixed order.
template <bool PL, bool SHARD>
__global__ __launch_bounds__(256) void k_linearize_lm_rows(BatchView V) {
  const int l = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int lane = threadIdx.x & 15;
  double acc[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) acc[q] = 0;
  bool live = l < V.nLr;
  if (live && !V.lm[V.lrow_graph[l]].lin) live = false;
  if (live) {
    const int li = V.lrow_lm[l];
    const double* lp = V.lmk + (size_t)li * 4;
    const int n = V.nEl;
    const bool is_point = !PL || V.lm_kind[li] == VT_POINT;
    for (int s = V.lslot_ptr[l] + lane; s < V.lslot_ptr[l + 1]; s += 16) {
      const int e = V.lslot_edge[s];
      const Pose Xi = load_pose(V.pose, V.el_p[e]);
      double err[3], Jl[9];
      if (is_point) {
        PointLin L;
        point_error(Xi, Vec3{lp[0], lp[1], lp[2]}, Vec3{V.el_z[0 * (size_t)n + e], V.el_z[1 * (size_t)n + e], V.el_z[2 * (size_t)n + e]}, L);
        err[0] = L.e[0]; err[1] = L.e[1]; err[2] = L.e[2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) Jl[a * 3 + c] = L.R.m[c * 3 + a];
      } else {
        const double* pj = V.pj + (size_t)e * kPjDoubles;   // from k_plane_jacobians: {e | J_l | ...}
#pragma unroll
        for (int q = 0; q < 3; ++q) err[q] = pj[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) Jl[q] = pj[3 + q];
      }
      double W[9];
      load_sym3(V.el_w, n, e, W);
      if (V.dcs_phi > 0) {
        const double r1 = dcs_rho1(V.dcs_phi, quad3(W, err));
#pragma unroll
        for (int q = 0; q < 9; ++q) W[q] *= r1;
      }
      if (SHARD) {
        const int g = V.lrow_graph[l];
        if (!(V.el_id[e] >= V.shard_lo[g] && V.el_id[e] < V.shard_hi[g])) {
#pragma unroll
          for (int q = 0; q < 9; ++q) W[q] = 0.0;
        }
      }
      double WJ[9], We[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        We[a] = W[a * 3 + 0] * err[0] + W[a * 3 + 1] * err[1] + W[a * 3 + 2] * err[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) WJ[a * 3 + c] = W[a * 3 + 0] * Jl[c] + W[a * 3 + 1] * Jl[3 + c] + W[a * 3 + 2] * Jl[6 + c];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[a * 3 + c] += Jl[a] * WJ[c] + Jl[3 + a] * WJ[3 + c] + Jl[6 + a] * WJ[6 + c];
        acc[9 + a] -= Jl[a] * We[0] + Jl[3 + a] * We[1] + Jl[6 + a] * We[2];
      }
    }
  }
  if (live && V.nEll > 0) {
    // point-point edges of this landmark: J = -I on the first vertex, +I on the second  ->  H_ll += Omega, b -/+= Omega e
    const size_t n = V.nEll;
    for (int s = V.llslot_ptr[l] + lane; s < V.llslot_ptr[l + 1]; s += 16) {
      const int2 rec = V.llslot_rec[s];
      const int k = rec.x;
      const double* pa = V.lmk + (size_t)V.ell_a[k] * 4;
      const double* pb = V.lmk + (size_t)V.ell_b[k] * 4;
      double err[3], W[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) err[r] = (pb[r] - pa[r]) - V.ell_z[r * n + k];
      load_sym3(V.ell_w, (int)n, k, W);
      if (SHARD) {
        const int g = V.lrow_graph[l];
        if (!(V.ell_id[k] >= V.shard_lo[g] && V.ell_id[k] < V.shard_hi[g])) {
#pragma unroll
          for (int q = 0; q < 9; ++q) W[q] = 0.0;
        }
      }
      const double sgn = rec.y ? -1.0 : 1.0;   // b = -J^T Omega e
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[a * 3 + c] += W[a * 3 + c];
        acc[9 + a] += sgn * (W[a * 3 + 0] * err[0] + W[a * 3 + 1] * err[1] + W[a * 3 + 2] * err[2]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    double v = acc[q];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    acc[q] = v;
  }
  if (live && lane == 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) V.Hll_diag[(size_t)l * 9 + q] = acc[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) V.bvec[(size_t)6 * V.nPr + (size_t)3 * l + q] = acc[9 + q];
  }
}

// landmark-landmark blocks of the point-point edges: block (a, b) = sum over the edges on that pair of J_a^T Omega J_b = -Omega
template <bool SHARD>
__global__ __launch_bounds__(64) void k_linearize_ll(BatchView V) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= V.nLL) return;
  const int e0 = V.llblk_ptr[i], e1 = V.llblk_ptr[i + 1];
  const int g = V.lrow_graph[V.lm_row[V.ell_a[V.llblk_edge[e0]]]];
  if (!V.lm[g].lin) return;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = e0; e < e1; ++e) {
    const int k = V.llblk_edge[e];
    if (SHARD && !(V.ell_id[k] >= V.shard_lo[g] && V.ell_id[k] < V.shard_hi[g])) continue;
    double W[9];
    load_sym3(V.ell_w, V.nEll, k, W);
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] -= W[q];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) V.Hll_off[(size_t)i * 9 + q] = acc[q];
}

// Further edges on an already-owned vertex pair (e.g. a repeated loop closure): their off-diagonal
// contributions are added serially, in edge order, after the owners have written the blocks.
__global__ void k_linearize_dups(BatchView V) {
  // the host orders the duplicate lists by block; thread t owns the run of entries that share its block
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int d = t; d < V.nDupEo; d = V.nDupEo) {
    if (d > 0 && (decode_blk(V.eo_blk[V.dup_eo[d - 1]]) >> 1) == (decode_blk(V.eo_blk[V.dup_eo[d]]) >> 1)) break;  // not a run head
    for (; d < V.nDupEo && (d == t || (decode_blk(V.eo_blk[V.dup_eo[d]]) >> 1) == (decode_blk(V.eo_blk[V.dup_eo[t]]) >> 1)); ++d) {
    const int k = V.dup_eo[d];
    const int pi = V.eo_i[k], pj = V.eo_j[k];
    const int gk = V.prow_graph[V.pose_row[pi]];
    if (!V.lm[gk].lin) continue;
    if (!(V.eo_id[k] >= V.shard_lo[gk] && V.eo_id[k] < V.shard_hi[gk])) continue;
    Se3Lin L;
    se3_error(load_pose(V.pose, pi), load_pose(V.pose, pj), load_meas_pose(V.eo_z, V.nEo, k), L);
    double Ji[36], Jj[36], W[36], WJ[36];
    se3_full_jacobians(L, Ji, Jj);
    load_sym6(V.eo_w, V.nEo, k, W);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double a = 0; for (int s = 0; s < 6; ++s) a += W[r * 6 + s] * Jj[s * 6 + c]; WJ[r * 6 + c] = a; }
    const int blk = decode_blk(V.eo_blk[k]);
    double* O = V.Hpp_off + (size_t)(blk >> 1) * 36;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
      double a = 0; for (int s = 0; s < 6; ++s) a += Ji[s * 6 + r] * WJ[s * 6 + c];
      O[(blk & 1) ? c * 6 + r : r * 6 + c] += a;
    }
    }
  }
  for (int d = t; d < V.nDupEl; d = V.nDupEl) {
    if (d > 0 && decode_blk(V.el_blk[V.dup_el[d - 1]]) == decode_blk(V.el_blk[V.dup_el[d]])) break;
    for (; d < V.nDupEl && (d == t || decode_blk(V.el_blk[V.dup_el[d]]) == decode_blk(V.el_blk[V.dup_el[t]])); ++d) {
    const int k = V.dup_el[d];
    const int pi = V.el_p[k], li = V.el_l[k];
    const int gk = V.prow_graph[V.pose_row[pi]];
    if (!V.lm[gk].lin) continue;
    if (!(V.el_id[k] >= V.shard_lo[gk] && V.el_id[k] < V.shard_hi[gk])) continue;
    const Pose Xi = load_pose(V.pose, pi);
    const double* lp = V.lmk + (size_t)li * 4;
    const int n = V.nEl;
    double Ji[18], Jl[9], W[9];
    if (V.lm_kind[li] == VT_POINT) {
      PointLin L;
      point_error(Xi, Vec3{lp[0], lp[1], lp[2]}, Vec3{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, L);
      point_jacobians(L, Ji, Jl);
    } else {
      plane_jacobians(Xi, Plane{{lp[0], lp[1], lp[2]}, lp[3]},
                      Plane{{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, V.el_z[3 * (size_t)n + k]}, Ji, Jl);
    }
    load_sym3(V.el_w, n, k, W);
    if (V.dcs_phi > 0) {
      double err[3];
      if (V.lm_kind[li] == VT_POINT) {
        PointLin L;
        point_error(Xi, Vec3{lp[0], lp[1], lp[2]}, Vec3{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, L);
        err[0] = L.e[0]; err[1] = L.e[1]; err[2] = L.e[2];
      } else {
        plane_error(Xi, Plane{{lp[0], lp[1], lp[2]}, lp[3]},
                    Plane{{V.el_z[0 * (size_t)n + k], V.el_z[1 * (size_t)n + k], V.el_z[2 * (size_t)n + k]}, V.el_z[3 * (size_t)n + k]}, err);
      }
      const double r1 = dcs_rho1(V.dcs_phi, quad3(W, err));
      for (int q = 0; q < 9; ++q) W[q] *= r1;
    }
    double WJl[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) WJl[r * 3 + c] = W[r * 3 + 0] * Jl[c] + W[r * 3 + 1] * Jl[3 + c] + W[r * 3 + 2] * Jl[6 + c];
    double* O = V.Hpl + (size_t)decode_blk(V.el_blk[k]) * 18;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) O[r * 3 + c] += Ji[r] * WJl[c] + Ji[6 + r] * WJl[3 + c] + Ji[12 + r] * WJl[6 + c];
    }
  }
}

// max diagonal entry per graph (lambda init = tau * max diag)
__global__ __launch_bounds__(kRowChunk) void k_maxdiag(BatchView V, double* __restrict__ part) {
  __shared__ double red[kRowChunk / 64];
  const int g = blockIdx.y;
  if (!V.lm[g].lin || V.lm[g].iter != 0) return;   // lambda is initialised from the first linearisation only
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  double d = 0;
  if (R.valid) d = fabs(R.is_pose ? V.Hpp_diag[(size_t)R.row * 36 + R.r * 7] : V.Hll_diag[(size_t)R.row * 9 + R.r * 4]);
  const double m = block_max<kRowChunk>(d, red);
  if (threadIdx.x == 0) part[(size_t)g * V.maxRowChunks + blockIdx.x] = m;
}

// One wave per graph, once per step.  A *step* is one damping trial: graphs flagged `lin` have just been re-linearised and
// start a new LM iteration (q = 0; lambda = tau * max diag on the very first one), the others are retrying the
// iteration they are in with the lambda k_lm_control raised.  The host never needs to know which is which.
__global__ void k_lm_begin_step(BatchView V, const double* __restrict__ part_maxdiag) {
  const int g = blockIdx.x;
  LmState& S = V.lm[g];
  if (!S.active) return;
  if (S.lin) {
    if (S.iter == 0) {
      const int n = row_chunks(V.seg[g]);
      double m = 0;
      for (int k = threadIdx.x; k < n; k += 64) m = fmax(m, part_maxdiag[(size_t)g * V.maxRowChunks + k]);
      for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
      if (threadIdx.x == 0) { S.max_diag = m; S.lambda = 1e-5 * m; S.nu = 2.0; }
    }
    if (threadIdx.x == 0) { S.q = 0; S.rho = 0; S.in_trial = 1; S.lin = 0; }
  }
  if (threadIdx.x == 0) S.accept = 0;
}

// initial chi2 -> cur_chi / chi_before.  Graphs with fewer than 10 edges are not optimised (graph_slam.cpp:184-186).
__global__ void k_lm_init(BatchView V, const double* __restrict__ part_chi, int min_edges) {
  const int g = blockIdx.x;
  LmState& S = V.lm[g];
  const GraphSeg sg = V.seg[g];
  const double c = wave_sum_partials(part_chi + (size_t)g * V.maxEdgeChunks, edge_chunks(sg));
  if (threadIdx.x == 0) {
    const bool few = sg.neo + sg.nel + sg.nell < min_edges;
    S.cur_chi = c; S.chi_before = c; S.iter = 0; S.trials = 0; S.pcg_iters = 0;
    S.status = few ? -5 : 0; S.active = few ? 0 : 1; S.lin = few ? 0 : 1;
    S.in_trial = 0; S.accept = 0; S.lambda = 0; S.nu = 2; S.rho = 0; S.q = 0; S.solve_failed = 0;
  }
}

// invert (D + lambda I) per block row -> block-Jacobi preconditioner
template <int N>
__device__ __forceinline__ void spd_inverse(double* A /* N*N row-major, in-place */) {
  // Cholesky A = L L^T (lower in place), then inverse via triangular inverse
  double L[N * N];
  for (int i = 0; i < N * N; ++i) L[i] = 0;
  for (int j = 0; j < N; ++j) {
    double d = A[j * N + j];
    for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k];
    d = sqrt(d);
    L[j * N + j] = d;
    for (int i = j + 1; i < N; ++i) {
      double s = A[i * N + j];
      for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
      L[i * N + j] = s / d;
    }
  }
  // Linv (lower)
  double Li[N * N];
  for (int i = 0; i < N * N; ++i) Li[i] = 0;
  for (int j = 0; j < N; ++j) {
    Li[j * N + j] = 1.0 / L[j * N + j];
    for (int i = j + 1; i < N; ++i) {
      double s = 0;
      for (int k = j; k < i; ++k) s -= L[i * N + k] * Li[k * N + j];
      Li[i * N + j] = s / L[i * N + i];
    }
  }
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      double s = 0;
      for (int k = (r > c ? r : c); k < N; ++k) s += Li[k * N + r] * Li[k * N + c];
      A[r * N + c] = s;
    }
}

__global__ __launch_bounds__(256) void k_precond(BatchView V) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < V.nPr) {
    const int g = V.prow_graph[t];
    if (!V.lm[g].in_trial) return;
    const double lam = V.lm[g].lambda;
    double A[36];
    for (int k = 0; k < 36; ++k) A[k] = V.Hpp_diag[(size_t)t * 36 + k];
    for (int k = 0; k < 6; ++k) A[k * 7] += lam;
    spd_inverse<6>(A);
    for (int k = 0; k < 36; ++k) V.Minv[(size_t)t * 36 + k] = A[k];
  } else if (t < V.nPr + V.nLr) {
    const int l = t - V.nPr;
    const int g = V.lrow_graph[l];
    if (!V.lm[g].in_trial) return;
    const double lam = V.lm[g].lambda;
    double A[9];
    for (int k = 0; k < 9; ++k) A[k] = V.Hll_diag[(size_t)l * 9 + k];
    for (int k = 0; k < 3; ++k) A[k * 4] += lam;
    spd_inverse<3>(A);
    for (int k = 0; k < 9; ++k) V.Minv[(size_t)V.nPr * 36 + (size_t)l * 9 + k] = A[k];
  }
}

// z = Minv * r for the block this scalar row belongs to (rvec staged in LDS by the caller)
__device__ __forceinline__ double apply_minv(const BatchView& V, const RowRef& R, const double* lds_r, int lds_base) {
  double z = 0;
  if (R.is_pose) {
    const double* M = V.Minv + (size_t)R.row * 36 + R.r * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) z += M[c] * lds_r[lds_base + c];
  } else {
    const double* M = V.Minv + (size_t)V.nPr * 36 + (size_t)R.row * 9 + R.r * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) z += M[c] * lds_r[lds_base + c];
  }
  return z;
}

// ---- Schur complement on the landmark block (solver 2, BASELINE.json north_star "Schur-complement + PCG") ------------------------
// PCG runs on the reduced pose system  S = (Hpp + lambda I) - Hpl (Hll + lambda I)^-1 Hlp  without ever forming it: for a vector v
// whose landmark part is  v_l = -(Hll + lambda I)^-1 Hlp v_p,  the pose rows of (H + lambda I) v are S v_p and its landmark rows
// vanish, so the full-space SpMV / update kernels are reused as they are.  One thread per landmark (its 3 x 3 inverse is the
// block-Jacobi preconditioner's, k_precond):
//   mode 0: v_l = -(Hll + lambda I)^-1 (Hlp v_p)
//   mode 1: v_l = -(Hll + lambda I)^-1 b_l and v_p = 0   (then the pose rows of (H + lambda I) v are  -Hpl (Hll + lambda I)^-1 b_l)
//   mode 2: v_l += (Hll + lambda I)^-1 b_l               (the landmarks' back-substitution:  x_l = (Hll + lambda I)^-1 (b_l - Hlp x_p))
__global__ __launch_bounds__(256) void k_schur_lm(BatchView V, double* v, int mode, int parity) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < V.nLr) {
    const int l = t, g = V.lrow_graph[l];
    if (!V.lm[g].in_trial) return;
    if (mode == 0 && V.pcg_done[parity * V.B + g]) return;
    const double* M = V.Minv + (size_t)V.nPr * 36 + (size_t)l * 9;
    const size_t base = (size_t)6 * V.nPr + (size_t)3 * l;
    double s[3] = {0, 0, 0};
    if (mode == 0) {
      const int arow = V.nPr + l;
      const double* Hbase = V.Hpp_diag;
      for (int a = V.adj_ptr[arow]; a < V.adj_ptr[arow + 1]; ++a) {   // a landmark row has pose neighbours only: blocks stored [pose][landmark]
        const double* Bk = Hbase + V.adj_blk[a];
        const double* pn = v + V.adj_x[a];
#pragma unroll
        for (int c = 0; c < 6; ++c) { s[0] += Bk[c * 3] * pn[c]; s[1] += Bk[c * 3 + 1] * pn[c]; s[2] += Bk[c * 3 + 2] * pn[c]; }
      }
    } else {
      s[0] = V.bvec[base]; s[1] = V.bvec[base + 1]; s[2] = V.bvec[base + 2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double m = M[r * 3] * s[0] + M[r * 3 + 1] * s[1] + M[r * 3 + 2] * s[2];
      if (mode == 2) v[base + r] += m; else v[base + r] = -m;
    }
  } else if (mode == 1 && t < V.nLr + V.nPr) {
    const int row = t - V.nLr;
    if (!V.lm[V.prow_graph[row]].in_trial) return;
#pragma unroll
    for (int c = 0; c < 6; ++c) v[(size_t)6 * row + c] = 0.0;
  }
}
__global__ void k_pcg_reset(BatchView V) {   // before the right-hand-side SpMV of the Schur path: flags of the previous solve are stale
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B) { const int d = V.lm[g].in_trial ? 0 : 1; V.pcg_done[g] = d; V.pcg_done[V.B + g] = d; }
}

// PCG start: x = 0, r = b, z = Minv r, p = z; partials of r.z and b.b  (schur: r_p = b_p + q_p with q from the right-hand-side SpMV, r_l = 0)
__global__ __launch_bounds__(kRowChunk) void k_pcg_init(BatchView V, int schur) {
  __shared__ double red[kRowChunk / 64];
  __shared__ double lr[kRowChunk];
  const int g = blockIdx.y;
  if (!V.lm[g].in_trial) return;
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  double rv = 0;
  if (R.valid) rv = !schur ? V.bvec[R.xoff] : (R.is_pose ? V.bvec[R.xoff] + V.q[R.xoff] : 0.0);
  lr[threadIdx.x] = rv;
  __syncthreads();
  double zv = 0;
  if (R.valid) {
    zv = apply_minv(V, R, lr, threadIdx.x - R.r);
    V.x[R.xoff] = 0; V.r[R.xoff] = rv; V.z[R.xoff] = zv; V.p[R.xoff] = zv;
  }
  const double s1 = block_sum<kRowChunk>(rv * zv, red);
  const double s2 = block_sum<kRowChunk>(rv * rv, red);
  if (threadIdx.x == 0) {
    V.part_a[(size_t)g * V.maxRowChunks + blockIdx.x] = s1;
    V.part_b[(size_t)g * V.maxRowChunks + blockIdx.x] = s2;
  }
}
__global__ void k_pcg_init2(BatchView V) {
  const int g = blockIdx.x;
  if (!V.lm[g].in_trial) { if (threadIdx.x == 0) { V.pcg_done[g] = 1; V.pcg_done[V.B + g] = 1; } return; }
  const int n = row_chunks(V.seg[g]);
  const double rz = wave_sum_partials(V.part_a + (size_t)g * V.maxRowChunks, n);
  const double bb = wave_sum_partials(V.part_b + (size_t)g * V.maxRowChunks, n);
  if (threadIdx.x == 0) {
    V.rz[g] = rz; V.rz[V.B + g] = rz; V.bb[g] = bb;
    V.pcg_fail[g] = 0;
    const int d = (bb == 0.0 || !(rz > 0)) ? 1 : 0;
    V.pcg_done[g] = d; V.pcg_done[V.B + g] = d;
    if (!(bb == 0.0) && !(rz > 0)) V.pcg_fail[g] = 1;
  }
}

// q = (H + lambda I) p, partial p.q      (block-sparse symmetric SpMV in gather form)
__global__ __launch_bounds__(kRowChunk) void k_spmv(BatchView V, int parity) {
  __shared__ double red[kRowChunk / 64];
  const int g = blockIdx.y;
  if (V.pcg_done[parity * V.B + g]) return;
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  double acc = 0, pv = 0;
  if (R.valid) {
    const double* __restrict__ p = V.p;
    pv = p[R.xoff];
    acc = V.lm[g].lambda * pv;
    int arow;
    if (R.is_pose) {
      const double* D = V.Hpp_diag + (size_t)R.row * 36 + R.r * 6;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc += D[c] * p[R.base + c];
      arow = R.row;
    } else {
      const double* D = V.Hll_diag + (size_t)R.row * 9 + R.r * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc += D[c] * p[R.base + c];
      arow = V.nPr + R.row;
    }
    const int a0 = V.adj_ptr[arow], a1 = V.adj_ptr[arow + 1];
    const double* Hbase = V.Hpp_diag;  // all H blocks live in one allocation; adj_blk is an offset from its start
    for (int a = a0; a < a1; ++a) {
      const double* Bk = Hbase + V.adj_blk[a];
      const double* pn = p + V.adj_x[a];
      const int fmt = V.adj_fmt[a];
      if (fmt == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += Bk[R.r * 6 + c] * pn[c];
      } else if (fmt == 1) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += Bk[c * 6 + R.r] * pn[c];
      } else if (fmt == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc += Bk[R.r * 3 + c] * pn[c];
      } else if (fmt == 4) {       // landmark-landmark block, this row first
#pragma unroll
        for (int c = 0; c < 3; ++c) acc += Bk[R.r * 3 + c] * pn[c];
      } else if (fmt == 5) {       // landmark-landmark block, this row second
#pragma unroll
        for (int c = 0; c < 3; ++c) acc += Bk[c * 3 + R.r] * pn[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += Bk[c * 3 + R.r] * pn[c];
      }
    }
    V.q[R.xoff] = acc;
  }
  const double s = block_sum<kRowChunk>(pv * acc, red);
  if (threadIdx.x == 0) V.part_a[(size_t)g * V.maxRowChunks + blockIdx.x] = s;
}

// alpha = rz / pq ; x += alpha p ; r -= alpha q ; z = Minv r ; partials of r.z, r.r
__global__ __launch_bounds__(kRowChunk) void k_pcg_update(BatchView V, int parity) {
  __shared__ double red[kRowChunk / 64];
  __shared__ double lr[kRowChunk];
  __shared__ double s_alpha;
  const int g = blockIdx.y;
  if (V.pcg_done[parity * V.B + g]) return;
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  if (threadIdx.x < 64) {
    const double pq = wave_sum_partials(V.part_a + (size_t)g * V.maxRowChunks, row_chunks(sg));
    if (threadIdx.x == 0) s_alpha = V.rz[parity * V.B + g] / pq;
  }
  __syncthreads();
  const double alpha = s_alpha;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  double rv = 0;
  if (R.valid) {
    V.x[R.xoff] += alpha * V.p[R.xoff];
    rv = V.r[R.xoff] - alpha * V.q[R.xoff];
    V.r[R.xoff] = rv;
  }
  lr[threadIdx.x] = rv;
  __syncthreads();
  double zv = 0;
  if (R.valid) { zv = apply_minv(V, R, lr, threadIdx.x - R.r); V.z[R.xoff] = zv; }
  const double s1 = block_sum<kRowChunk>(rv * zv, red);
  const double s2 = block_sum<kRowChunk>(rv * rv, red);
  if (threadIdx.x == 0) {
    V.part_b[(size_t)g * V.maxRowChunks + blockIdx.x] = s1;
    V.part_c[(size_t)g * V.maxRowChunks + blockIdx.x] = s2;  // (part_a is still being read by other blocks)
  }
}

// beta = rz_new / rz ; p = z + beta p ; convergence bookkeeping (flags double-buffered by parity so
// that no block of a launch can observe a flag written by another block of the same launch)
__global__ __launch_bounds__(kRowChunk) void k_pcg_pupdate(BatchView V, int parity, double tol2, int max_iters) {
  __shared__ double s_beta;
  const int g = blockIdx.y;
  int* done_in = V.pcg_done + parity * V.B;
  int* done_out = V.pcg_done + (parity ^ 1) * V.B;
  if (done_in[g]) {
    if (blockIdx.x == 0 && threadIdx.x == 0) done_out[g] = 1;
    return;
  }
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  if (threadIdx.x < 64) {
    const int n = row_chunks(sg);
    const double rzn = wave_sum_partials(V.part_b + (size_t)g * V.maxRowChunks, n);
    const double rr = wave_sum_partials(V.part_c + (size_t)g * V.maxRowChunks, n);
    if (threadIdx.x == 0) {
      const double rzo = V.rz[parity * V.B + g];
      s_beta = rzn / rzo;
      if (blockIdx.x == 0) {
        V.rz[(parity ^ 1) * V.B + g] = rzn;
        LmState& S = V.lm[g];
        S.pcg_iters += 1;
        int d = 0;
        if (!(rr > tol2 * V.bb[g])) d = 1;                 // converged (or NaN)
        if (!isfinite(rr) || !isfinite(rzn) || !(rzn > 0)) { d = 1; if (!(rr <= tol2 * V.bb[g])) V.pcg_fail[g] = 1; }
        done_out[g] = d;
      }
    }
  }
  __syncthreads();
  const double beta = s_beta;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  if (R.valid) V.p[R.xoff] = V.z[R.xoff] + beta * V.p[R.xoff];
}

__global__ void k_pcg_alldone(BatchView V, int parity) {
  // one workgroup; flags[1] = 1 iff every graph's PCG is finished
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  for (int g = threadIdx.x; g < V.B; g += blockDim.x)
    if (!V.pcg_done[parity * V.B + g]) s_any = 1;
  __syncthreads();
  if (threadIdx.x == 0) V.flags[1] = s_any ? 0 : 1;
}

// trial estimate = current [+] dx  (VertexSE3 / VertexPointXYZ / VertexPlane oplus)
__global__ __launch_bounds__(256) void k_oplus(BatchView V, const double* __restrict__ dx) { oplus_row(V, blockIdx.x * blockDim.x + threadIdx.x, dx); }

// partial sums of dx . (lambda dx + b)   (denominator of the LM gain ratio)
__global__ __launch_bounds__(kRowChunk) void k_scale(BatchView V, const double* __restrict__ dx) {
  __shared__ double red[kRowChunk / 64];
  const int g = blockIdx.y;
  if (!V.lm[g].in_trial) return;
  const GraphSeg sg = V.seg[g];
  if (blockIdx.x * kRowChunk >= sg.nprow * 6 + sg.nlrow * 3) return;
  const RowRef R = row_ref(V, sg, blockIdx.x * kRowChunk + threadIdx.x);
  double v = 0;
  if (R.valid) { const double d = dx[R.xoff]; v = d * (V.lm[g].lambda * d + V.bvec[R.xoff]); }
  const double s = block_sum<kRowChunk>(v, red);
  if (threadIdx.x == 0) V.part_a[(size_t)g * V.maxRowChunks + blockIdx.x] = s;
}

// g2o OptimizationAlgorithmLevenberg accept / reject, one wave per graph
__global__ void k_lm_control(BatchView V, const double* __restrict__ part_chi, int max_iters) {
  const int g = blockIdx.x;
  LmState& S = V.lm[g];
  if (!S.in_trial) return;
  const double tchi = wave_sum_partials(part_chi + (size_t)g * V.maxEdgeChunks, edge_chunks(V.seg[g]));
  const double sc = wave_sum_partials(V.part_a + (size_t)g * V.maxRowChunks, row_chunks(V.seg[g]));
  if (threadIdx.x != 0) return;
  lm_control_apply(S, tchi, sc, V.pcg_fail[g], max_iters);
}

// an accepted trial becomes the estimate (commit_row for every block row): four lanes per pose (64 B with its pad), two per landmark (32 B),
// 16 bytes each -- a wave moves 1 KB of consecutive bytes per instruction (one thread per row with seven 8-byte loads 64 B apart ran at
// 1.7 TB/s: 0.21 ms of the 512-graph step)
__global__ __launch_bounds__(256) void k_commit(BatchView V) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 4 * V.nPr) {
    const int r = t >> 2, q = t & 3;
    if (!V.lm[V.prow_graph[r]].accept) return;
    const size_t pi = (size_t)V.prow_pose[r];
    reinterpret_cast<D2*>(V.pose + pi * 8)[q] = reinterpret_cast<const D2*>(V.pose_trial + pi * 8)[q];
  } else if (t < 4 * V.nPr + 2 * V.nLr) {
    const int u = t - 4 * V.nPr, l = u >> 1, q = u & 1;
    if (!V.lm[V.lrow_graph[l]].accept) return;
    const size_t li = (size_t)V.lrow_lm[l];
    reinterpret_cast<D2*>(V.lmk + li * 4)[q] = reinterpret_cast<const D2*>(V.lmk_trial + li * 4)[q];
  }
}
__global__ void k_set_trial_all(BatchView V, double lambda) {  // used by the solve() hook
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B) { V.lm[g].in_trial = 1; V.lm[g].lambda = lambda; V.lm[g].active = 1; }
}

}  // namespace sslam

// =============================================================================================
// Host engine
// =============================================================================================
namespace sslam {

// RCCL is bound with dlopen when the edge-sharded mode is first used, not at link time: a host program that already carries its
// own copy (PyTorch ships one) must end up with ONE librccl in the process, and the single-GPU path must not depend on it at all.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllReduce && GetErrorString; }
};
static RcclApi& rccl_api() {
  static RcclApi api;
  if (!api.lib) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);   // an already loaded copy with this soname is reused
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    }
  }
  return api;
}
void batch_comm_destroy(void* comm) { if (comm && rccl_api().ok()) (void)rccl_api().CommDestroy((ncclComm_t)comm); }

std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}

static inline uint64_t pair_key(int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

// Compile host graphs into the device-resident batch layout (g2o initializeOptimization +
// BlockSolver::buildStructure analogue; symbolic work only).
static int batch_build(Batch& b, bool host_only = false) {
  if (!host_only) {
    SSLAM_HIP_TRY(hipSetDevice(b.device));
    if (!b.stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  }
  const int B = (int)b.graphs.size();
  b.seg.assign(B, GraphSeg{});
  b.v2pose.assign(B, {}); b.v2lm.assign(B, {});
  b.pose_row.clear(); b.lm_row.clear(); b.prow_pose.clear(); b.lrow_lm.clear();
  b.ppoff.clear(); b.plblk.clear(); b.llblk.clear(); b.pose_vertex.clear(); b.lm_vertex.clear();
  std::vector<int> prow_graph, lrow_graph;
  std::vector<unsigned char> lm_kind;
  std::vector<int> eo_i, eo_j, eo_blk, el_p, el_l, el_blk;
  std::vector<int> eo_src, el_src;  // (graph-local edge id) for SoA fill
  std::vector<int> eo_g, el_g;
  std::vector<int> ell_a, ell_b, ell_src, ell_g, ell_blk;   // point-point edges (g2o::EdgePointXYZ)
  std::vector<int> shard_lo(b.graphs.size(), 0), shard_hi(b.graphs.size(), 0x7fffffff);
  int maxRow = 1, maxEdge = 1;
  for (int g = 0; g < B; ++g) {
    const HostGraph& G = *b.graphs[g];
    std::vector<int> hidx;
    hessian_indices(G, hidx);
    GraphSeg& sg = b.seg[g];
    sg.pose0 = (int)b.pose_row.size(); sg.lm0 = (int)b.lm_row.size();
    sg.prow0 = (int)b.prow_pose.size(); sg.lrow0 = (int)b.lrow_lm.size();
    sg.eo0 = (int)eo_i.size(); sg.el0 = (int)el_p.size(); sg.ell0 = (int)ell_a.size();
    auto& vp = b.v2pose[g]; auto& vl = b.v2lm[g];
    vp.assign(G.nv(), -1); vl.assign(G.nv(), -1);
    for (int v = 0; v < G.nv(); ++v) {
      if (G.vtype[v] == VT_SE3) {
        vp[v] = (int)b.pose_row.size();
        b.pose_vertex.push_back(v);
        if (hidx[v] >= 0) { b.pose_row.push_back((int)b.prow_pose.size()); b.prow_pose.push_back(vp[v]); prow_graph.push_back(g); }
        else b.pose_row.push_back(-1);
      } else {
        vl[v] = (int)b.lm_row.size();
        b.lm_vertex.push_back(v);
        lm_kind.push_back((unsigned char)G.vtype[v]);
        if (hidx[v] >= 0) { b.lm_row.push_back((int)b.lrow_lm.size()); b.lrow_lm.push_back(vl[v]); lrow_graph.push_back(g); }
        else b.lm_row.push_back(-1);
      }
    }
    sg.npose = (int)b.pose_row.size() - sg.pose0; sg.nlm = (int)b.lm_row.size() - sg.lm0;
    sg.nprow = (int)b.prow_pose.size() - sg.prow0; sg.nlrow = (int)b.lrow_lm.size() - sg.lrow0;
    for (int k = 0; k < G.ne(); ++k) {
      if (G.etype[k] == ET_SE3) { eo_i.push_back(vp[G.evi[k]]); eo_j.push_back(vp[G.evj[k]]); eo_src.push_back(k); eo_g.push_back(g); }
      else if (G.etype[k] == ET_POINT_POINT) { ell_a.push_back(vl[G.evi[k]]); ell_b.push_back(vl[G.evj[k]]); ell_src.push_back(k); ell_g.push_back(g); }
      else { el_p.push_back(vp[G.evi[k]]); el_l.push_back(vl[G.evj[k]]); el_src.push_back(k); el_g.push_back(g); }
    }
    sg.neo = (int)eo_i.size() - sg.eo0; sg.nel = (int)el_p.size() - sg.el0; sg.nell = (int)ell_a.size() - sg.ell0;
    maxRow = std::max(maxRow, (sg.nprow * 6 + sg.nlrow * 3 + kRowChunk - 1) / kRowChunk);
    maxEdge = std::max(maxEdge, (sg.neo + sg.nel + sg.nell + kEdgeChunk - 1) / kEdgeChunk);
  }
  const int nPr = (int)b.prow_pose.size(), nLr = (int)b.lrow_lm.size();
  const int nEo = (int)eo_i.size(), nEl = (int)el_p.size(), nEll = (int)ell_a.size();
  // unique off-diagonal blocks
  std::unordered_map<uint64_t, int> ppmap, plmap, llmap;
  std::vector<std::vector<int>> llblk_edges;                 // per landmark-landmark block: its edges
  std::vector<std::vector<std::pair<int, int>>> llslots(nLr);   // per landmark row: {edge, side}
  ell_blk.assign(nEll, -1);
  for (int k = 0; k < nEll; ++k) {
    const int ra = b.lm_row[ell_a[k]], rb = b.lm_row[ell_b[k]];
    if (ra >= 0) llslots[ra].push_back({k, 0});
    if (rb >= 0) llslots[rb].push_back({k, 1});
    if (ra < 0 || rb < 0 || ra == rb) continue;
    const int a = std::min(ra, rb), c = std::max(ra, rb);
    auto it = llmap.find(pair_key(a, c));
    int idx;
    if (it == llmap.end()) { idx = (int)b.llblk.size(); llmap.emplace(pair_key(a, c), idx); b.llblk.push_back({a, c}); llblk_edges.push_back({}); }
    else idx = it->second;
    llblk_edges[idx].push_back(k);
    ell_blk[k] = idx;
  }
  eo_blk.assign(nEo, -1); el_blk.assign(nEl, -1);
  for (int k = 0; k < nEo; ++k) {
    const int ri = b.pose_row[eo_i[k]], rj = b.pose_row[eo_j[k]];
    if (ri < 0 || rj < 0 || ri == rj) continue;
    const int a = std::min(ri, rj), c = std::max(ri, rj);
    auto it = ppmap.find(pair_key(a, c));
    int idx;
    if (it == ppmap.end()) { idx = (int)b.ppoff.size(); ppmap.emplace(pair_key(a, c), idx); b.ppoff.push_back({a, c}); }
    else idx = it->second;
    eo_blk[k] = idx * 2 + (ri > rj ? 1 : 0);
  }
  for (int k = 0; k < nEl; ++k) {
    const int rp = b.pose_row[el_p[k]], rl = b.lm_row[el_l[k]];
    if (rp < 0 || rl < 0) continue;
    auto it = plmap.find(pair_key(rp, rl));
    int idx;
    if (it == plmap.end()) { idx = (int)b.plblk.size(); plmap.emplace(pair_key(rp, rl), idx); b.plblk.push_back({rp, rl}); }
    else idx = it->second;
    el_blk[k] = idx;
  }
  // (row, incident edge) slots for the gather-form Jacobian build; duplicates of an off-diagonal
  // block (two edges on the same vertex pair) force the atomic variant
  std::vector<std::vector<std::pair<int, unsigned char>>> pslots(nPr);
  std::vector<std::vector<int>> lslots(nLr);
  {
    std::vector<char> seen_pp(b.ppoff.size(), 0), seen_pl(b.plblk.size(), 0);
    b.has_duplicate_blocks = false;
    b.dup_eo.clear(); b.dup_el.clear();
    b.has_planes = false;
    for (unsigned char k : lm_kind) if (k == VT_PLANE) b.has_planes = true;
    for (int k = 0; k < nEo; ++k) {
      const int ri = b.pose_row[eo_i[k]], rj = b.pose_row[eo_j[k]];
      if (ri >= 0) pslots[ri].push_back({k, 0});
      if (rj >= 0) pslots[rj].push_back({k, 1});
      if (eo_blk[k] >= 0) {   // a later edge on an already-owned block: not the owner (code <= -2)
        if (seen_pp[eo_blk[k] >> 1]) { b.has_duplicate_blocks = true; b.dup_eo.push_back(k); eo_blk[k] = -2 - eo_blk[k]; }
        else seen_pp[eo_blk[k] >> 1] = 1;
      }
    }
    for (int k = 0; k < nEl; ++k) {
      const int rp = b.pose_row[el_p[k]], rl = b.lm_row[el_l[k]];
      if (rp >= 0) pslots[rp].push_back({k, 2});
      if (rl >= 0) lslots[rl].push_back(k);
      if (el_blk[k] >= 0) {
        if (seen_pl[el_blk[k]]) { b.has_duplicate_blocks = true; b.dup_el.push_back(k); el_blk[k] = -2 - el_blk[k]; }
        else seen_pl[el_blk[k]] = 1;
      }
    }
  }
  std::stable_sort(b.dup_eo.begin(), b.dup_eo.end(), [&](int x, int y) { return ((-2 - eo_blk[x]) >> 1) < ((-2 - eo_blk[y]) >> 1); });
  std::stable_sort(b.dup_el.begin(), b.dup_el.end(), [&](int x, int y) { return (-2 - el_blk[x]) < (-2 - el_blk[y]); });
  std::vector<int> pslot_ptr(nPr + 1, 0), pslot_edge, lslot_ptr(nLr + 1, 0), lslot_edge, tile_row0, tile_row1;
  std::vector<unsigned char> pslot_kind;
  b.max_row_slots = 0;
  // record = {edge, kind (0: EdgeSE3 seen from its i side, 1: from its j side, 2: landmark edge), vertex i | pose, vertex j | landmark}
  std::vector<int4> pslot_rec;
  for (int r = 0; r < nPr; ++r) {
    for (auto& s : pslots[r]) {
      pslot_edge.push_back(s.first); pslot_kind.push_back(s.second);
      if (s.second < 2) pslot_rec.push_back(make_int4(s.first, s.second, eo_i[s.first], eo_j[s.first]));
      else pslot_rec.push_back(make_int4(s.first, 2, el_p[s.first], el_l[s.first]));
    }
    pslot_ptr[r + 1] = (int)pslot_edge.size();
    b.max_row_slots = std::max(b.max_row_slots, (int)pslots[r].size());
  }
  // thread -> row of the pose-row kernel: within a graph, rows with the same numbers of EdgeSE3 and landmark slots next to each other.  The
  // lanes of a wave walk their rows' slots in step; a chain pose has two EdgeSE3 slots, the ~200 endpoints of the loop closures of an L
  // graph three or four, and spread over the graph's 78 waves they made nearly every wave run the EdgeSE3 arithmetic a third and fourth
  // time for one or two lanes.  Sorted, those rows fill three waves of their own (2.07 -> 2.02 ms per 512-graph build).
  std::vector<int> prow_perm(nPr);
  for (int r = 0; r < nPr; ++r) prow_perm[r] = r;
  {
    std::vector<int> nse3(nPr, 0);
    for (int r = 0; r < nPr; ++r) for (auto& sl : pslots[r]) nse3[r] += sl.second < 2 ? 1 : 0;
    for (int g = 0; g < B; ++g) {
      const GraphSeg& sg = b.seg[g];
      std::stable_sort(prow_perm.begin() + sg.prow0, prow_perm.begin() + sg.prow0 + sg.nprow, [&](int x, int y) {
        if (nse3[x] != nse3[y]) return nse3[x] < nse3[y];
        return pslots[x].size() < pslots[y].size();
      });
    }
  }
  for (int r = 0; r < nLr; ++r) {
    for (int k : lslots[r]) lslot_edge.push_back(k);
    lslot_ptr[r + 1] = (int)lslot_edge.size();
  }
  for (int g = 0; g < B; ++g) {   // tiles never span graphs
    const GraphSeg& sg = b.seg[g];
    int r = sg.prow0;
    const int rend = sg.prow0 + sg.nprow;
    while (r < rend) {
      int r1 = r, ns = 0;
      while (r1 < rend && r1 - r < 32 && ns + (pslot_ptr[r1 + 1] - pslot_ptr[r1]) <= kTileSlots) { ns += pslot_ptr[r1 + 1] - pslot_ptr[r1]; ++r1; }
      if (r1 == r) ++r1;  // a single row with more than kTileSlots slots: atomic variant is used instead
      tile_row0.push_back(r); tile_row1.push_back(r1);
      r = r1;
    }
  }
  const int nPP = (int)b.ppoff.size(), nPL = (int)b.plblk.size(), nLL = (int)b.llblk.size();
  b.hll_base = (int64_t)nPr * 36;
  b.hpp_off_base = b.hll_base + (((int64_t)nLr * 9 + 1) & ~(int64_t)1);   // even: the 6-wide blocks behind it stay 16-byte aligned
  b.hpl_base = b.hpp_off_base + (int64_t)nPP * 36;
  b.hll_off_base = b.hpl_base + (int64_t)nPL * 18;
  const int64_t h_total = b.hll_off_base + (int64_t)nLL * 9;
  if (h_total >= (int64_t)1 << 31) return set_error(SSLAM_ERR_INVALID, "batch too large: H has %lld doubles (int32 block offsets)", (long long)h_total);
  if (host_only) {   // symbolic structure only (plan introspection on a box without a GPU)
    memset(&b.V, 0, sizeof b.V);
    b.V.B = B; b.V.nPr = nPr; b.V.nLr = nLr; b.V.h_total = h_total;
    return 0;
  }
  // adjacency (rows: pose rows then landmark rows)
  std::vector<std::vector<std::array<int, 3>>> adj(nPr + nLr);
  for (int i = 0; i < nPP; ++i) {
    const int a = b.ppoff[i].first, c = b.ppoff[i].second;
    const int off = (int)(b.hpp_off_base + (int64_t)i * 36);
    adj[a].push_back({off, 6 * c, 0});
    adj[c].push_back({off, 6 * a, 1});
  }
  for (int i = 0; i < nPL; ++i) {
    const int rp = b.plblk[i].first, rl = b.plblk[i].second;
    const int off = (int)(b.hpl_base + (int64_t)i * 18);
    adj[rp].push_back({off, 6 * nPr + 3 * rl, 2});
    adj[nPr + rl].push_back({off, 6 * rp, 3});
  }
  for (int i = 0; i < nLL; ++i) {
    const int a = b.llblk[i].first, c = b.llblk[i].second;
    const int off = (int)(b.hll_off_base + (int64_t)i * 9);
    adj[nPr + a].push_back({off, 6 * nPr + 3 * c, 4});
    adj[nPr + c].push_back({off, 6 * nPr + 3 * a, 5});
  }
  std::vector<int> adj_ptr(nPr + nLr + 1, 0), adj_blk, adj_x;
  std::vector<unsigned char> adj_fmt;
  for (int r = 0; r < nPr + nLr; ++r) {
    std::sort(adj[r].begin(), adj[r].end(), [](const std::array<int, 3>& x, const std::array<int, 3>& y) { return x[1] < y[1]; });
    for (auto& a : adj[r]) { adj_blk.push_back(a[0]); adj_x.push_back(a[1]); adj_fmt.push_back((unsigned char)a[2]); }
    adj_ptr[r + 1] = (int)adj_blk.size();
  }
  // edge SoA payloads
  std::vector<double> eo_z((size_t)7 * nEo), eo_w((size_t)21 * nEo), el_z((size_t)4 * nEl), el_w((size_t)6 * nEl);
  for (int k = 0; k < nEo; ++k) {
    const HostGraph& G = *b.graphs[eo_g[k]];
    const int s = eo_src[k];
    for (int c = 0; c < 7; ++c) eo_z[(size_t)c * nEo + k] = G.meas[(size_t)s * 7 + c];
    int q = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) eo_w[(size_t)(q++) * nEo + k] = G.info[(size_t)s * 36 + r * 6 + c];
  }
  for (int k = 0; k < nEl; ++k) {
    const HostGraph& G = *b.graphs[el_g[k]];
    const int s = el_src[k];
    for (int c = 0; c < 4; ++c) el_z[(size_t)c * nEl + k] = G.meas[(size_t)s * 7 + c];
    int q = 0;
    for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) el_w[(size_t)(q++) * nEl + k] = G.info[(size_t)s * 36 + r * 3 + c];
  }
  std::vector<double> ell_z((size_t)3 * nEll), ell_w((size_t)6 * nEll);
  for (int k = 0; k < nEll; ++k) {
    const HostGraph& G = *b.graphs[ell_g[k]];
    const int s = ell_src[k];
    for (int c = 0; c < 3; ++c) ell_z[(size_t)c * nEll + k] = G.meas[(size_t)s * 7 + c];
    int q = 0;
    for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) ell_w[(size_t)(q++) * nEll + k] = G.info[(size_t)s * 36 + r * 3 + c];
  }
  std::vector<int> llslot_ptr(nLr + 1, 0), llblk_ptr(nLL + 1, 0), llblk_edge;
  std::vector<int2> llslot_rec;
  for (int r = 0; r < nLr; ++r) { for (auto& q : llslots[r]) llslot_rec.push_back(make_int2(q.first, q.second)); llslot_ptr[r + 1] = (int)llslot_rec.size(); }
  for (int i = 0; i < nLL; ++i) { for (int k : llblk_edges[i]) llblk_edge.push_back(k); llblk_ptr[i + 1] = (int)llblk_edge.size(); }
  // ---- device allocation
  BatchView& V = b.V;
  memset(&V, 0, sizeof V);
  V.B = B; V.nPr = nPr; V.nLr = nLr; V.nPose = (int)b.pose_row.size(); V.nLm = (int)b.lm_row.size();
  V.nEo = nEo; V.nEl = nEl; V.nEll = nEll; V.nLL = nLL; V.maxRowChunks = maxRow; V.maxEdgeChunks = maxEdge;
  V.h_total = h_total;
  V.dcs_phi = b.graphs[0]->opt.dcs_phi;
  int rc;
#define UP(vec, field) if ((rc = dev_upload(b, vec, (std::remove_const<std::remove_pointer<decltype(V.field)>::type>::type**)&V.field)) != 0) return rc
  UP(b.seg, seg); UP(b.pose_row, pose_row); UP(b.lm_row, lm_row); UP(b.prow_pose, prow_pose); UP(b.lrow_lm, lrow_lm);
  UP(prow_graph, prow_graph); UP(lrow_graph, lrow_graph); UP(lm_kind, lm_kind); UP(prow_perm, prow_perm);
  UP(eo_i, eo_i); UP(eo_j, eo_j); UP(eo_z, eo_z); UP(eo_w, eo_w); UP(eo_blk, eo_blk);
  UP(el_p, el_p); UP(el_l, el_l); UP(el_z, el_z); UP(el_w, el_w); UP(el_blk, el_blk);
  UP(adj_ptr, adj_ptr); UP(adj_blk, adj_blk); UP(adj_x, adj_x); UP(adj_fmt, adj_fmt);
  UP(tile_row0, tile_row0); UP(tile_row1, tile_row1); UP(pslot_ptr, pslot_ptr); UP(pslot_edge, pslot_edge); UP(pslot_kind, pslot_kind);
  UP(lslot_ptr, lslot_ptr); UP(lslot_edge, lslot_edge);
  UP(b.dup_eo, dup_eo); UP(b.dup_el, dup_el); UP(pslot_rec, pslot_rec);
  UP(eo_src, eo_id); UP(el_src, el_id); UP(shard_lo, shard_lo); UP(shard_hi, shard_hi);
  UP(ell_a, ell_a); UP(ell_b, ell_b); UP(ell_z, ell_z); UP(ell_w, ell_w); UP(ell_src, ell_id);
  UP(llslot_ptr, llslot_ptr); UP(llslot_rec, llslot_rec); UP(llblk_ptr, llblk_ptr); UP(llblk_edge, llblk_edge);
  V.nDupEo = (int)b.dup_eo.size(); V.nDupEl = (int)b.dup_el.size();
  V.nTiles = (int)tile_row0.size();
#undef UP
  double* H = nullptr;
  const size_t dim = (size_t)6 * nPr + (size_t)3 * nLr;
  const size_t h_even = ((size_t)h_total + 1) & ~(size_t)1;   // b right behind H (16-byte aligned): [H || b] is one all-reduce buffer
  if ((rc = dev_alloc(b, h_even + dim, &H))) return rc;
  V.Hpp_diag = H; V.Hll_diag = H + b.hll_base; V.Hpp_off = H + b.hpp_off_base; V.Hpl = H + b.hpl_base; V.Hll_off = H + b.hll_off_base;
  V.bvec = H + h_even;
  b.hb_doubles = (int64_t)(h_even + dim);
  V.pj = nullptr;
  if (b.has_planes && (rc = dev_alloc(b, (size_t)std::max(nEl, 1) * kPjDoubles, &V.pj))) return rc;
  if ((rc = dev_alloc(b, (size_t)V.nPose * 8, &V.pose))) return rc;
  if ((rc = dev_alloc(b, (size_t)V.nPose * 8, &V.pose_trial))) return rc;
  if ((rc = dev_alloc(b, (size_t)V.nLm * 4, &V.lmk))) return rc;
  if ((rc = dev_alloc(b, (size_t)V.nLm * 4, &V.lmk_trial))) return rc;
  if ((rc = dev_alloc(b, dim, &V.x))) return rc;
  if ((rc = dev_alloc(b, dim, &V.r))) return rc;
  if ((rc = dev_alloc(b, dim, &V.z))) return rc;
  if ((rc = dev_alloc(b, dim, &V.p))) return rc;
  if ((rc = dev_alloc(b, dim, &V.q))) return rc;
  if ((rc = dev_alloc(b, (size_t)nPr * 36 + (size_t)nLr * 9, &V.Minv))) return rc;
  if ((rc = dev_alloc(b, (size_t)B * maxRow, &V.part_a))) return rc;
  if ((rc = dev_alloc(b, (size_t)B * maxRow, &V.part_b))) return rc;
  if ((rc = dev_alloc(b, (size_t)B * maxRow, &V.part_c))) return rc;
  if ((rc = dev_alloc(b, (size_t)B * maxRow, &b.d_part_m))) return rc;
  if ((rc = dev_alloc(b, (size_t)B * maxEdge, &b.d_part_e))) return rc;
  if ((rc = dev_alloc(b, (size_t)2 * B, &V.rz))) return rc;
  if ((rc = dev_alloc(b, (size_t)B, &V.bb))) return rc;
  if ((rc = dev_alloc(b, (size_t)2 * B, &V.pcg_done))) return rc;
  if ((rc = dev_alloc(b, (size_t)B, &V.pcg_fail))) return rc;
  if ((rc = dev_alloc(b, (size_t)4, &V.flags))) return rc;
  if ((rc = dev_alloc(b, (size_t)B, &V.lm))) return rc;
  if (b.arena && b.arena->flush(b.stream)) return set_error(SSLAM_ERR_HIP, "upload of the batch tables failed");
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  b.versions.resize(B);
  for (int g = 0; g < B; ++g) b.versions[g] = b.graphs[g]->structure_version;
  b.uploaded = false;
  return 0;
}

static int batch_upload_estimates(Batch& b) {
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  const size_t np = (size_t)b.V.nPose * 8, nl = (size_t)b.V.nLm * 4;
  std::vector<double> fallback;
  double* stage = reinterpret_cast<double*>(b.pin->get((np + nl) * sizeof(double)));   // page-locked: four asynchronous copies, no staged blits
  if (!stage) { fallback.assign(np + nl, 0.0); stage = fallback.data(); }
  std::fill(stage, stage + np + nl, 0.0);
  double* pose = stage; double* lmk = stage + np;
  for (size_t g = 0; g < b.graphs.size(); ++g) {
    const HostGraph& G = *b.graphs[g];
    for (int v = 0; v < G.nv(); ++v) {
      const double* e = &G.est[(size_t)v * 7];
      if (G.vtype[v] == VT_SE3) { double* o = &pose[(size_t)b.v2pose[g][v] * 8]; for (int k = 0; k < 7; ++k) o[k] = e[k]; }
      else { double* o = &lmk[(size_t)b.v2lm[g][v] * 4]; for (int k = 0; k < (G.vtype[v] == VT_POINT ? 3 : 4); ++k) o[k] = e[k]; }
    }
  }
  if (np) {
    SSLAM_HIP_TRY(hipMemcpyAsync(b.V.pose, pose, np * 8, hipMemcpyHostToDevice, b.stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(b.V.pose_trial, pose, np * 8, hipMemcpyHostToDevice, b.stream));
  }
  if (nl) {
    SSLAM_HIP_TRY(hipMemcpyAsync(b.V.lmk, lmk, nl * 8, hipMemcpyHostToDevice, b.stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(b.V.lmk_trial, lmk, nl * 8, hipMemcpyHostToDevice, b.stream));
  }
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  b.uploaded = true;
  return 0;
}

static int batch_download_estimates(Batch& b) {
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  const size_t np = (size_t)b.V.nPose * 8, nl = (size_t)b.V.nLm * 4;
  std::vector<double> fallback;
  double* stage = reinterpret_cast<double*>(b.pin->get((np + nl) * sizeof(double)));
  if (!stage) { fallback.assign(np + nl, 0.0); stage = fallback.data(); }
  double* pose = stage; double* lmk = stage + np;
  if (np) SSLAM_HIP_TRY(hipMemcpyAsync(pose, b.V.pose, np * 8, hipMemcpyDeviceToHost, b.stream));
  if (nl) SSLAM_HIP_TRY(hipMemcpyAsync(lmk, b.V.lmk, nl * 8, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  for (size_t g = 0; g < b.graphs.size(); ++g) {
    HostGraph& G = *b.graphs[g];
    for (int v = 0; v < G.nv(); ++v) {
      double* e = &G.est[(size_t)v * 7];
      if (G.vtype[v] == VT_SE3) { const double* o = &pose[(size_t)b.v2pose[g][v] * 8]; for (int k = 0; k < 7; ++k) e[k] = o[k]; }
      else { const double* o = &lmk[(size_t)b.v2lm[g][v] * 4]; for (int k = 0; k < (G.vtype[v] == VT_POINT ? 3 : 4); ++k) e[k] = o[k]; }
    }
  }
  return 0;
}

// a few device bytes -> host through the batch's page-locked staging buffer; synchronises the stream
static int read_device(Batch& b, void* host, const void* dev, size_t bytes) {
  char* stage = b.pin->get(bytes);
  SSLAM_HIP_TRY(hipMemcpyAsync(stage ? (void*)stage : host, dev, bytes, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  if (stage) memcpy(host, stage, bytes);
  return 0;
}

static inline dim3 row_grid(const Batch& b) { return dim3(b.V.maxRowChunks, b.V.B); }
static inline dim3 edge_grid(const Batch& b) { return dim3(b.V.maxEdgeChunks, b.V.B); }
static inline int vert_blocks(const Batch& b) { return std::max(1, (b.V.nPr + b.V.nLr + 255) / 256); }

static int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "launch %s: %s", what, hipGetErrorString(e));
  return 0;
}

// BlockSolver::buildSystem for the active graphs of the batch
static int batch_linearize(Batch& b) {
  ScopedTimer t(b, "linearize");
  b.V.dcs_phi = b.graphs[0]->opt.dcs_phi;   // read live like pcg_tol / solver: an option set after the batch was built must not be ignored
  const BatchView& V = b.V;
  const int nblk = (V.nPr + kRowThreads - 1) / kRowThreads;
#define SSLAM_LAUNCH_LIN(PLV, SHV)                                                                                                    \
  {                                                                                                                                   \
    if (PLV && V.nEl > 0) {                                                                                                           \
      const long long bA = ((long long)V.nEl * 14 + 255) / 256, bB = ((long long)V.nEl * 6 + 255) / 256;                                \
      hipLaunchKernelGGL(k_plane_jacobians, dim3((unsigned)(bA + bB)), dim3(256), 0, b.stream, V, (int)bA);                              \
    }                                                                                                                                 \
    if (V.nPr > 0) hipLaunchKernelGGL((k_linearize_rowthread<PLV, SHV>), dim3(nblk), dim3(kRowThreads), 0, b.stream, V);               \
    if (V.nLr > 0) hipLaunchKernelGGL((k_linearize_lm_rows<PLV, SHV>), dim3((V.nLr + 15) / 16), dim3(256), 0, b.stream, V);            \
    if (V.nLL > 0) hipLaunchKernelGGL((k_linearize_ll<SHV>), dim3((V.nLL + 63) / 64), dim3(64), 0, b.stream, V);                       \
  }
  // (Round 4 could run the landmark-row kernel on a second stream next to the pose-row kernel: +4 % on one stream, -24 % inside a stream
  // group, where every part forked and joined its own second stream at every step -- removed in round 5.)
  if (b.sharded) {
    // edge-sharded mode: the rank-partial system is built in its own buffer and summed OUT OF PLACE into [H || b].  A graph that does not
    // re-linearise in this step (a rejected trial being retried, a finished graph) keeps its old partial system there, so the sum over
    // ranks is again the system it already had: the all-reduce is idempotent for it.  (Summing in place would multiply the H of such a
    // graph by `world` on every step: the damping of a retried trial would fall instead of rising -- round-2 ADVICE.)
    BatchView Vp = V;
    Vp.Hpp_diag = b.d_hb_part; Vp.Hll_diag = b.d_hb_part + b.hll_base; Vp.Hpp_off = b.d_hb_part + b.hpp_off_base;
    Vp.Hpl = b.d_hb_part + b.hpl_base; Vp.Hll_off = b.d_hb_part + b.hll_off_base; Vp.bvec = b.d_hb_part + (V.bvec - V.Hpp_diag);
    {
      const BatchView& V = Vp;
      if (b.has_planes) SSLAM_LAUNCH_LIN(true, true) else SSLAM_LAUNCH_LIN(false, true)
      if (V.nDupEo + V.nDupEl > 0) hipLaunchKernelGGL(k_linearize_dups, dim3((std::max(V.nDupEo, V.nDupEl) + 63) / 64), dim3(64), 0, b.stream, V);
    }
    if (b.comm) {
      // the partial [H || b] arrays of all ranks -> their sum on every rank: ONE all-reduce over xGMI (RCCL), in stream order;
      // everything after it (solve, update, chi2, LM control) runs replicated and bit-identical on all ranks
      const ncclResult_t r = rccl_api().AllReduce(b.d_hb_part, V.Hpp_diag, (size_t)b.hb_doubles, ncclDouble, ncclSum, (ncclComm_t)b.comm, b.stream);
      if (r != ncclSuccess) return set_error(SSLAM_ERR_HIP, "ncclAllReduce of the normal equations: %s", rccl_api().GetErrorString(r));
      b.allreduce_calls++;
    } else {   // no communicator (single-device parity hook): the partial system is what the caller reads
      SSLAM_HIP_TRY(hipMemcpyAsync(V.Hpp_diag, b.d_hb_part, (size_t)b.hb_doubles * sizeof(double), hipMemcpyDeviceToDevice, b.stream));
    }
  } else {
    if (b.has_planes) SSLAM_LAUNCH_LIN(true, false) else SSLAM_LAUNCH_LIN(false, false)
    if (V.nDupEo + V.nDupEl > 0) hipLaunchKernelGGL(k_linearize_dups, dim3((std::max(V.nDupEo, V.nDupEl) + 63) / 64), dim3(64), 0, b.stream, V);
  }
#undef SSLAM_LAUNCH_LIN
  return launch_check("linearize");
}

static int pcg_solve(Batch& b) {
  const Options& opt = b.graphs[0]->opt;
  const BatchView& V = b.V;
  if (opt.solver == 2 && V.nLL > 0)
    return set_error(SSLAM_ERR_UNSUPPORTED, "solver 2 (Schur complement on the landmark block) needs a block-diagonal landmark block: the graph holds point-point edges");
  { ScopedTimer t(b, "precond");
    hipLaunchKernelGGL(k_precond, dim3(vert_blocks(b)), dim3(256), 0, b.stream, V); }
  const bool schur = opt.solver == 2;
  const dim3 lm_grid((V.nLr + V.nPr + 255) / 256);
  if (schur) {   // reduced right-hand side  b_p - Hpl (Hll + lambda I)^-1 b_l  through one SpMV
    hipLaunchKernelGGL(k_pcg_reset, dim3((V.B + 63) / 64), dim3(64), 0, b.stream, V);
    hipLaunchKernelGGL(k_schur_lm, lm_grid, dim3(256), 0, b.stream, V, V.p, 1, 0);
    hipLaunchKernelGGL(k_spmv, row_grid(b), dim3(kRowChunk), 0, b.stream, V, 0);
  }
  hipLaunchKernelGGL(k_pcg_init, row_grid(b), dim3(kRowChunk), 0, b.stream, V, schur ? 1 : 0);
  hipLaunchKernelGGL(k_pcg_init2, dim3(V.B), dim3(64), 0, b.stream, V);
  const double tol2 = opt.pcg_tol * opt.pcg_tol;
  int it = 0;
  const int check_every = 32;
  while (it < opt.pcg_max_iters) {
    for (int k = 0; k < check_every && it < opt.pcg_max_iters; ++k, ++it) {
      const int parity = it & 1;
      if (schur) hipLaunchKernelGGL(k_schur_lm, lm_grid, dim3(256), 0, b.stream, V, V.p, 0, parity);
      { ScopedTimer t(b, "spmv");
        hipLaunchKernelGGL(k_spmv, row_grid(b), dim3(kRowChunk), 0, b.stream, V, parity); }
      { ScopedTimer t(b, "pcg_update");
        hipLaunchKernelGGL(k_pcg_update, row_grid(b), dim3(kRowChunk), 0, b.stream, V, parity);
        hipLaunchKernelGGL(k_pcg_pupdate, row_grid(b), dim3(kRowChunk), 0, b.stream, V, parity, tol2, opt.pcg_max_iters); }
    }
    hipLaunchKernelGGL(k_pcg_alldone, dim3(1), dim3(256), 0, b.stream, V, it & 1);
    int flag = 0;
    if (int rc2 = read_device(b, &flag, V.flags + 1, sizeof(int))) return rc2;
    b.harvest();
    if (flag) break;
  }
  if (schur) hipLaunchKernelGGL(k_schur_lm, lm_grid, dim3(256), 0, b.stream, V, V.x, 2, 0);
  return launch_check("pcg");
}

static int batch_solve(Batch& b) {
  // (H + lambda I) dx = b for every graph with in_trial set; result in V.x
  if (b.graphs[0]->opt.solver == 0 || b.graphs[0]->opt.solver == 2) return pcg_solve(b);
  int rc;
  if (!b.chol) {
    static const bool timing = getenv("SSLAM_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = chol_plan_build(b))) return rc;
    if (timing) fprintf(stderr, "[timing] cholesky plan build %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  if (chol_plan_flow(b)) return chol_solve_flow(b);
  if ((rc = chol_factor_and_forward(b))) return rc;
  return chol_backward(b);
}

static int batch_chi2(Batch& b, const double* pose, const double* lmk, int mask_mode) {
  ScopedTimer t(b, "chi2");
  b.V.dcs_phi = b.graphs[0]->opt.dcs_phi;
  hipLaunchKernelGGL(k_chi2, edge_grid(b), dim3(kEdgeChunk), 0, b.stream, b.V, pose, lmk, mask_mode, b.d_part_e);
  return launch_check("chi2");
}

constexpr int kStepChunk = 8;   // LM steps enqueued between two looks at the per-graph state
// the per-graph LM states -> host, through the page-locked staging buffer (synchronises the stream)
static int read_lm_state(Batch& b, std::vector<LmState>& st) { return read_device(b, st.data(), b.V.lm, sizeof(LmState) * (size_t)b.V.B); }

static int batch_optimize(Batch& b, int max_iters, sslam_opt_stats* out) {
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  const auto t0 = std::chrono::steady_clock::now();
  const BatchView& V = b.V;
  int rc;
  if (!b.uploaded && (rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, V.pose, V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(V.B), dim3(64), 0, b.stream, V, b.d_part_e, 10);
  std::vector<LmState> st(V.B);
  // The LM control flow (accept / reject, retry with a larger lambda, terminate) lives on the device: the host enqueues
  // generic steps in chunks and only looks at the per-graph state between chunks -- no synchronisation per trial.
  // Every graph needs at least (max_iters - iter) more steps; rejected trials add steps, which later chunks supply.
  // Small batches whose plan runs factor + solves in one dependency-driven launch (k_chol_flow): the rest of a damping trial in two more
  // launches (k_lm_begin_small / k_lm_end_small) -- same sums, same order, same results as the stand-alone kernels
  bool flow_steps = false;
  {
    if (b.graphs[0]->opt.fused && b.graphs[0]->opt.solver == 1 && !b.sharded && !b.profiling) {
      if (!b.chol && (rc = chol_plan_build(b))) return rc;
      flow_steps = chol_plan_flow(b);
    }
  }
  const bool spec_steps = flow_steps && chol_spec_mode(b) && chol_plan_spec(b);
  long long budget = 10LL * std::max(max_iters, 0) + 8;    // hard bound: <= 10 trials per iteration
  int need = max_iters;
  if ((rc = chol_set_active(b, nullptr))) return rc;
  struct CompactGuard { Batch& b; ~CompactGuard() { (void)chol_set_active(b, nullptr); } } compact_guard{b};   // every exit leaves the launches sized for all graphs
  std::vector<char> act(V.B, 1);
  int n_act = V.B;
  static const bool chunk_timing = getenv("SSLAM_TIMING") != nullptr;
  bool looked = false;   // st holds the state after the last enqueued step
  while (need > 0 && budget > 0) {
    const int chunk = (int)std::min<long long>(std::min(need, kStepChunk), budget);
    const auto tq0 = std::chrono::steady_clock::now();
    for (int sidx = 0; sidx < chunk; ++sidx) {
      if ((rc = batch_linearize(b))) return rc;
      if (spec_steps) { if ((rc = chol_lm_step_spec(b, max_iters))) return rc; continue; }
      if (flow_steps) { if ((rc = chol_lm_step_flow(b, max_iters))) return rc; continue; }
      hipLaunchKernelGGL(k_maxdiag, row_grid(b), dim3(kRowChunk), 0, b.stream, V, b.d_part_m);
      hipLaunchKernelGGL(k_lm_begin_step, dim3(V.B), dim3(64), 0, b.stream, V, b.d_part_m);
      if ((rc = batch_solve(b))) return rc;
      { ScopedTimer t(b, "oplus");
        hipLaunchKernelGGL(k_oplus, dim3(vert_blocks(b)), dim3(256), 0, b.stream, V, V.x); }
      if ((rc = batch_chi2(b, V.pose_trial, V.lmk_trial, 1))) return rc;
      hipLaunchKernelGGL(k_scale, row_grid(b), dim3(kRowChunk), 0, b.stream, V, V.x);
      hipLaunchKernelGGL(k_lm_control, dim3(V.B), dim3(64), 0, b.stream, V, b.d_part_e, max_iters);
      hipLaunchKernelGGL(k_commit, dim3(std::max(1, (4 * V.nPr + 2 * V.nLr + 255) / 256)), dim3(256), 0, b.stream, V);
    }
    budget -= chunk;
    const auto tq1 = std::chrono::steady_clock::now();
    if ((rc = read_lm_state(b, st))) return rc;
    looked = true;
    if (chunk_timing && V.B == 1)
      fprintf(stderr, "[timing] LM chunk: %d steps enqueued in %.3f ms, waited %.3f ms more; iteration %d trials %d active %d\n", chunk,
              std::chrono::duration<double, std::milli>(tq1 - tq0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq1).count(),
              st[0].iter, st[0].trials, st[0].active);
    b.harvest();
    need = 0;
    int na = 0;
    for (int g = 0; g < V.B; ++g) {
      act[g] = st[g].active ? 1 : 0; na += act[g];
      if (!st[g].active) continue;
      // a graph in a streak of rejected trials ends its iteration -- at convergence, its optimisation -- after at most 10 - q more of
      // them: enqueue exactly those (g2o's LM ends every optimisation of the orchestrator with ten rejected trials; a full chunk
      // behind each look cost 4-6 idle steps per tick).  An accepted trial in between: the next look supplies more steps.
      // (small batches only: on a large batch every look is a host round trip for all graphs, and short chunks in the endgame cost the
      // 512-graph stream group 12 % -- 25.0k vs 28.5k iterations/s)
      const int streak_left = (V.B < 8 && st[g].in_trial && st[g].q > 0) ? std::max(1, 10 - st[g].q) : kStepChunk;
      need = std::max(need, std::min(max_iters - st[g].iter, streak_left));
    }
    // the graphs that are still iterating only ever shrink: once half of the batch is done, the factor / solve launches are sized
    // for the rest (a retry by three graphs then costs three graphs' pieces, not the dispatch of everybody's)
    static const bool timing = getenv("SSLAM_TIMING") != nullptr;
    if (timing && V.B > 1) fprintf(stderr, "[timing] LM chunk of %d steps done: %d of %d graphs still active, %d more iterations needed\n", chunk, na, V.B, need);
    if (need > 0 && na < n_act && 2 * na <= V.B) {
      const auto tc = std::chrono::steady_clock::now();
      if ((rc = chol_set_active(b, &act))) return rc;
      n_act = na;
      if (timing) fprintf(stderr, "[timing] launches resized for %d active graphs in %.3f ms\n", na, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count());
    }
  }
  if ((rc = chol_set_active(b, nullptr))) return rc;
  if (!looked && (rc = read_lm_state(b, st))) return rc;   // (nothing was enqueued after the loop's last look)
  b.harvest();
  if ((rc = launch_check("optimize"))) return rc;
  if ((rc = chol_flow_check(b))) return rc;
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int worst = 0;
  for (int g = 0; g < V.B; ++g) {
    sslam_opt_stats& o = out[g];
    memset(&o, 0, sizeof o);
    o.iterations = st[g].iter; o.trials = st[g].trials; o.status = st[g].status;
    o.chi2_before = st[g].chi_before; o.chi2_after = st[g].cur_chi; o.lambda = st[g].lambda;
    o.seconds = secs; o.solver_iterations = st[g].pcg_iters;
    if (st[g].status != SSLAM_ERR_TOO_FEW_EDGES && !std::isfinite(st[g].cur_chi)) { o.status = SSLAM_ERR_NUMERIC; worst = SSLAM_ERR_NUMERIC; }
  }
  if (worst) return set_error(worst, "non-finite chi2 after optimisation");
  return 0;
}

}  // namespace sslam

// =============================================================================================
// C-ABI
// =============================================================================================
using namespace sslam;

struct sslam_graph {
  HostGraph g;
  DevArena arena;                // device memory of the batch below, kept across structure rebuilds (declared first: destroyed last)
  PinnedScratch pin;             // page-locked staging of the batch's small copies, kept across rebuilds as well
  hipStream_t stream = nullptr;  // the handle's stream, handed to every batch it builds (a stream create / destroy pair per tick costs more
                                 // than optimising a small graph)
  std::unique_ptr<Batch> batch;  // batch of one, rebuilt when the structure changes
  bool linearized = false;
  ~sslam_graph() {
    batch.reset();               // synchronises the stream
    if (stream) { (void)hipSetDevice(g.device); (void)hipStreamSynchronize(stream); persist_forget_stream(g.device, stream); (void)hipStreamDestroy(stream); }
  }
};
struct sslam_batch {
  Batch b;
  // Stream group (sslam_batch_create_streams): the graphs split into contiguous parts, every part a batch of its own with its own HIP
  // stream, optimised from its own host thread.  A batch marches batch-synchronously through the depths of its graphs' elimination trees
  // and through its LM trial rounds; parts that are out of phase with one another fill the chip while one of them is at the narrow top
  // of its trees or in the LM endgame with a few graphs left.  `b` is unused in a group handle.
  std::vector<sslam_batch*> parts;
  std::vector<int> part0;      // first graph of every part, then n
  ~sslam_batch() { for (sslam_batch* p : parts) delete p; }
};

// run f(part, index) for every part of a group handle, one host thread per part (f = nullptr-safe single batches never come here);
// the first failing part's status and message are handed to the calling thread
template <class F>
static int for_each_part(sslam_batch* h, bool threaded, F f) {
  const int K = (int)h->parts.size();
  std::vector<int> rcs(K, 0);
  std::vector<std::string> errs(K);
  if (threaded) {
    std::vector<std::thread> th;
    for (int k = 0; k < K; ++k)
      th.emplace_back([&, k] { rcs[k] = f(h->parts[k], k); if (rcs[k]) errs[k] = last_error_ref(); });
    for (auto& t : th) t.join();
  } else {
    for (int k = 0; k < K; ++k) { rcs[k] = f(h->parts[k], k); if (rcs[k]) errs[k] = last_error_ref(); }
  }
  for (int k = 0; k < K; ++k) if (rcs[k]) return set_error(rcs[k], "part %d of the stream group: %s", k, errs[k].c_str());
  return 0;
}

static int ensure_batch(sslam_graph* h) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible (%s); the product has no CPU fallback", hipGetErrorString(e));
  if (h->g.device < 0 || h->g.device >= n) return set_error(SSLAM_ERR_NO_DEVICE, "device %d out of range (%d visible)", h->g.device, n);
  if (!h->batch || h->batch->versions.empty() || h->batch->versions[0] != h->g.structure_version) {
    static const bool timing = getenv("SSLAM_TIMING") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    h->batch.reset();                              // the old batch synchronises its stream; its arrays belong to the arena
    SSLAM_HIP_TRY(hipSetDevice(h->g.device));
    h->arena.reset();
    if (timing) fprintf(stderr, "[timing] rebuild: release of the old batch + arena reset %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count());
    h->batch.reset(new Batch());
    h->batch->arena = &h->arena;
    h->batch->pin = &h->pin;
    h->batch->device = h->g.device;
    if (!h->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->batch->stream = h->stream; h->batch->own_stream = false;
    h->batch->graphs = {&h->g};
    int rc = batch_build(*h->batch);
    if (rc) { h->batch.reset(); return rc; }
    h->linearized = false;
  }
  return 0;
}

extern "C" {

const char* sslam_last_error(void) { return last_error_ref().c_str(); }

int sslam_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// page-locked host memory for buffers that travel to the device asynchronously (the clouds of sslam_seg_submit_batch): a caller that
// does not link HIP itself gets it here
void* sslam_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    set_error(SSLAM_ERR_NO_DEVICE, "hipHostMalloc of %zu bytes failed (no HIP device?)", bytes);
    return nullptr;
  }
  return p;
}
void sslam_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

sslam_graph* sslam_graph_create(int device) {
  sslam_graph* h = new sslam_graph();
  h->g.device = device;
  return h;
}
void sslam_graph_destroy(sslam_graph* g) { delete g; }

static int add_vertex(sslam_graph* h, int type, const double* est, int n, int fixed) {
  if (!h || !est) return set_error(SSLAM_ERR_INVALID, "null argument");
  HostGraph& G = h->g;
  const int id = G.nv();
  G.vtype.push_back(type);
  G.vfixed.push_back(fixed);
  for (int k = 0; k < 7; ++k) G.est.push_back(k < n ? est[k] : 0.0);
  G.structure_version++;
  return id;
}
int sslam_graph_add_vertex_se3(sslam_graph* h, const double t_q[7], int fixed) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null graph");
  if (fixed < 0) fixed = (h->g.nv() == 0) ? 1 : 0;  // graph_slam.cpp:109-111
  return add_vertex(h, VT_SE3, t_q, 7, fixed ? 1 : 0);
}
int sslam_graph_add_vertex_point(sslam_graph* h, const double p[3]) { return add_vertex(h, VT_POINT, p, 3, 0); }
int sslam_graph_add_vertex_plane(sslam_graph* h, const double n_d[4]) {
  if (!n_d) return set_error(SSLAM_ERR_INVALID, "null argument");
  const double nn = std::sqrt(n_d[0] * n_d[0] + n_d[1] * n_d[1] + n_d[2] * n_d[2]);
  if (!(nn > 0)) return set_error(SSLAM_ERR_INVALID, "plane normal has zero length");
  const double p[4] = {n_d[0] / nn, n_d[1] / nn, n_d[2] / nn, n_d[3] / nn};  // Plane3D::fromVector normalises
  return add_vertex(h, VT_PLANE, p, 4, 0);
}

static int add_edge(sslam_graph* h, int type, int i, int j, const double* z, int nz, const double* info, int d) {
  if (!h || !z || !info) return set_error(SSLAM_ERR_INVALID, "null argument");
  HostGraph& G = h->g;
  if (i < 0 || j < 0 || i >= G.nv() || j >= G.nv() || i == j) return set_error(SSLAM_ERR_INVALID, "edge vertex ids (%d,%d) invalid", i, j);
  if (G.vtype[i] != (type == ET_POINT_POINT ? VT_POINT : VT_SE3)) return set_error(SSLAM_ERR_INVALID, "vertex %d has the wrong type for this edge", i);
  const int want = type == ET_SE3 ? VT_SE3 : ((type == ET_SE3_POINT || type == ET_POINT_POINT) ? VT_POINT : VT_PLANE);
  if (G.vtype[j] != want) return set_error(SSLAM_ERR_INVALID, "vertex %d has the wrong type for this edge", j);
  // only the upper triangle of the information matrix travels to the device: an asymmetric one would be symmetrised silently
  double amax = 0;
  for (int k = 0; k < d * d; ++k) { if (!std::isfinite(info[k])) return set_error(SSLAM_ERR_INVALID, "information matrix has a non-finite entry"); amax = std::max(amax, std::fabs(info[k])); }
  for (int r = 0; r < d; ++r)
    for (int c = r + 1; c < d; ++c)
      if (std::fabs(info[r * d + c] - info[c * d + r]) > 1e-9 * amax)
        return set_error(SSLAM_ERR_INVALID, "information matrix is not symmetric (entries (%d,%d) and (%d,%d) differ)", r, c, c, r);
  const int id = G.ne();
  G.etype.push_back(type); G.evi.push_back(i); G.evj.push_back(j);
  for (int k = 0; k < 7; ++k) G.meas.push_back(k < nz ? z[k] : 0.0);
  for (int k = 0; k < 36; ++k) G.info.push_back(k < d * d ? info[k] : 0.0);
  G.structure_version++;
  return id;
}
int sslam_graph_add_edge_se3(sslam_graph* h, int i, int j, const double z[7], const double info[36]) {
  return add_edge(h, ET_SE3, i, j, z, 7, info, 6);
}
int sslam_graph_add_edge_se3_point(sslam_graph* h, int i, int l, const double z[3], const double info[9]) {
  return add_edge(h, ET_SE3_POINT, i, l, z, 3, info, 3);
}
int sslam_graph_add_edge_se3_plane(sslam_graph* h, int i, int l, const double z[4], const double info[9]) {
  if (!z) return set_error(SSLAM_ERR_INVALID, "null argument");
  const double nn = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  if (!(nn > 0)) return set_error(SSLAM_ERR_INVALID, "plane normal has zero length");
  const double p[4] = {z[0] / nn, z[1] / nn, z[2] / nn, z[3] / nn};
  return add_edge(h, ET_SE3_PLANE, i, l, p, 4, info, 3);
}

int sslam_graph_add_edge_point_point(sslam_graph* h, int l1, int l2, const double z[3], const double info[9]) {
  return add_edge(h, ET_POINT_POINT, l1, l2, z, 3, info, 3);
}

int sslam_graph_num_vertices(const sslam_graph* h) { return h ? h->g.nv() : SSLAM_ERR_INVALID; }
int sslam_graph_num_edges(const sslam_graph* h) { return h ? h->g.ne() : SSLAM_ERR_INVALID; }

int sslam_graph_get_vertex(const sslam_graph* h, int id, double* out) {
  if (!h || !out || id < 0 || id >= h->g.nv()) return set_error(SSLAM_ERR_INVALID, "bad vertex id %d", id);
  const int n = vertex_est_len(h->g.vtype[id]);
  for (int k = 0; k < n; ++k) out[k] = h->g.est[(size_t)id * 7 + k];
  return n;
}
int sslam_graph_set_vertex(sslam_graph* h, int id, const double* in) {
  if (!h || !in || id < 0 || id >= h->g.nv()) return set_error(SSLAM_ERR_INVALID, "bad vertex id %d", id);
  const int n = vertex_est_len(h->g.vtype[id]);
  for (int k = 0; k < n; ++k) h->g.est[(size_t)id * 7 + k] = in[k];
  if (h->batch) h->batch->uploaded = false;
  h->linearized = false;
  return n;
}
int sslam_graph_hessian_index(sslam_graph* h, int id) {
  if (!h || id < 0 || id >= h->g.nv()) return set_error(SSLAM_ERR_INVALID, "bad vertex id %d", id);
  std::vector<int> hidx;
  hessian_indices(h->g, hidx);
  return hidx[id];
}

int sslam_graph_set_option(sslam_graph* h, const char* key, double value) {
  if (!h || !key) return set_error(SSLAM_ERR_INVALID, "null argument");
  Options& o = h->g.opt;
  const std::string k(key);
  if (k == "solver") { if (value != 0 && value != 1 && value != 2) return set_error(SSLAM_ERR_INVALID, "solver: 0 block-Jacobi PCG, 1 sparse block Cholesky, 2 Schur complement on the landmarks + PCG (3, the window plan of round 3, was removed in round 5: 2-2.7x slower than 1)"); o.solver = (int)value; }
  else if (k == "robust_kernel_dcs") { if (!(value >= 0)) return set_error(SSLAM_ERR_INVALID, "robust_kernel_dcs: phi >= 0 (0 = no kernel)"); o.dcs_phi = value; if (h->batch) h->batch->V.dcs_phi = value; h->linearized = false; }
  else if (k == "fused_small_graph") o.fused = value != 0;
  else if (k == "speculative_trials") o.speculative = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
  else if (k == "pcg_tol") o.pcg_tol = value;
  else if (k == "pcg_max_iters") o.pcg_max_iters = (int)value;
  else if (k == "deterministic") { if (value == 0) return set_error(SSLAM_ERR_UNSUPPORTED, "the Jacobian build is always deterministic (gather form); the FP64-atomics variant was removed"); }
  else return set_error(SSLAM_ERR_INVALID, "unknown option '%s'", key);
  return 0;
}

int sslam_graph_optimize(sslam_graph* h, int max_iters, sslam_opt_stats* out) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null graph");
  sslam_opt_stats local;
  if (!out) out = &local;
  memset(out, 0, sizeof *out);
  if (h->g.ne() < 10) {  // graph_slam.cpp:184-186
    out->status = SSLAM_ERR_TOO_FEW_EDGES;
    return set_error(SSLAM_ERR_TOO_FEW_EDGES, "graph has %d edges (< 10): not optimised", h->g.ne());
  }
  static const bool timing = getenv("SSLAM_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  const bool reuse = h->batch && !h->batch->versions.empty() && h->batch->versions[0] == h->g.structure_version;   // ensure_batch rebuilds otherwise
  if (h->batch) h->batch->plan_build_ms = 0;   // (a call that failed after a rebuild must not leak its plan time into this one: round-5 ADVICE)
  int rc = ensure_batch(h);
  if (rc) { out->status = rc; return rc; }
  const double t1 = now();
  if ((rc = batch_upload_estimates(*h->batch))) return rc;
  const double t2 = now();
  if ((rc = batch_optimize(*h->batch, max_iters, out))) return rc;
  const double t3 = now();
  // what a structure change cost the host: batch tables (ensure_batch) + symbolic factorisation; exactly 0 when the structure was reused
  out->host_plan_us = (reuse && h->batch->plan_build_ms == 0) ? 0 : (int)std::lround(1e3 * ((reuse ? 0.0 : t1 - t0) + h->batch->plan_build_ms));
  h->batch->plan_build_ms = 0;
  h->linearized = false;
  rc = batch_download_estimates(*h->batch);
  if (timing) fprintf(stderr, "[timing] optimize: vertices %d edges %d | structure %.3f upload %.3f LM %.3f (iterations %d trials %d) download %.3f ms\n",
                      h->g.nv(), h->g.ne(), t1 - t0, t2 - t1, t3 - t2, out->iterations, out->trials, now() - t3);
  return rc;
}

int sslam_graph_chi2(sslam_graph* h, double* chi2) {
  if (!h || !chi2) return set_error(SSLAM_ERR_INVALID, "null argument");
  int rc = ensure_batch(h);
  if (rc) return rc;
  Batch& b = *h->batch;
  if ((rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, b.V.pose, b.V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(b.V.B), dim3(64), 0, b.stream, b.V, b.d_part_e, 0);
  LmState s;
  if ((rc = read_device(b, &s, b.V.lm, sizeof s))) return rc;
  *chi2 = s.cur_chi;
  return 0;
}

static int do_linearize(sslam_graph* h) {
  int rc = ensure_batch(h);
  if (rc) return rc;
  Batch& b = *h->batch;
  if ((rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, b.V.pose, b.V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(b.V.B), dim3(64), 0, b.stream, b.V, b.d_part_e, 0);
  if ((rc = batch_linearize(b))) return rc;
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  b.harvest();
  h->linearized = true;
  return 0;
}

int sslam_graph_linearize(sslam_graph* h, int* dim, int64_t* nnz_upper, int32_t* rows, int32_t* cols, double* vals, double* bout) {
  if (!h || !dim || !nnz_upper) return set_error(SSLAM_ERR_INVALID, "null argument");
  int rc = ensure_batch(h);
  if (rc) return rc;
  Batch& b = *h->batch;
  std::vector<int> hidx;
  const int n = hessian_indices(h->g, hidx);
  *dim = n;
  const int nPr = b.V.nPr, nLr = b.V.nLr;
  const int64_t nnz = (int64_t)nPr * 21 + (int64_t)nLr * 6 + (int64_t)b.ppoff.size() * 36 + (int64_t)b.plblk.size() * 18 + (int64_t)b.llblk.size() * 9;
  *nnz_upper = nnz;
  if (!rows || !cols || !vals || !bout) return 0;
  if ((rc = do_linearize(h))) return rc;
  std::vector<double> H((size_t)b.V.h_total), bv((size_t)6 * nPr + (size_t)3 * nLr);
  SSLAM_HIP_TRY(hipMemcpy(H.data(), b.V.Hpp_diag, H.size() * 8, hipMemcpyDeviceToHost));
  SSLAM_HIP_TRY(hipMemcpy(bv.data(), b.V.bvec, bv.size() * 8, hipMemcpyDeviceToHost));
  auto prow_off = [&](int r) { return hidx[b.pose_vertex[b.prow_pose[r]]]; };
  auto lrow_off = [&](int r) { return hidx[b.lm_vertex[b.lrow_lm[r]]]; };
  int64_t q = 0;
  auto emit = [&](int R, int C, double v) { if (R > C) std::swap(R, C); rows[q] = R; cols[q] = C; vals[q] = v; ++q; };
  for (int r = 0; r < nPr; ++r) {
    const int o = prow_off(r);
    for (int a = 0; a < 6; ++a) { for (int c = a; c < 6; ++c) emit(o + a, o + c, H[(size_t)r * 36 + a * 6 + c]); bout[o + a] = bv[(size_t)6 * r + a]; }
  }
  for (int r = 0; r < nLr; ++r) {
    const int o = lrow_off(r);
    for (int a = 0; a < 3; ++a) { for (int c = a; c < 3; ++c) emit(o + a, o + c, H[(size_t)b.hll_base + (size_t)r * 9 + a * 3 + c]); bout[o + a] = bv[(size_t)6 * nPr + (size_t)3 * r + a]; }
  }
  for (size_t i = 0; i < b.ppoff.size(); ++i) {
    const int oa = prow_off(b.ppoff[i].first), oc = prow_off(b.ppoff[i].second);
    for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) emit(oa + a, oc + c, H[(size_t)b.hpp_off_base + i * 36 + a * 6 + c]);
  }
  for (size_t i = 0; i < b.plblk.size(); ++i) {
    const int op = prow_off(b.plblk[i].first), ol = lrow_off(b.plblk[i].second);
    for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) emit(op + a, ol + c, H[(size_t)b.hpl_base + i * 18 + a * 3 + c]);
  }
  for (size_t i = 0; i < b.llblk.size(); ++i) {   // emit() orders (row, column) into the upper triangle; the block is symmetric (-Omega)
    const int oa = lrow_off(b.llblk[i].first), oc = lrow_off(b.llblk[i].second);
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) {
      const double v = H[(size_t)b.hll_off_base + i * 9 + a * 3 + c];
      if (oa < oc) emit(oa + a, oc + c, v); else emit(oc + c, oa + a, v);
    }
  }
  return 0;
}

// internal <-> g2o ordering of the unknown vector
static void to_g2o_order(const sslam_graph* h, const std::vector<double>& xin, double* xout) {
  const Batch& b = *h->batch;
  std::vector<int> hidx;
  hessian_indices(h->g, hidx);
  for (int r = 0; r < b.V.nPr; ++r) { const int o = hidx[b.pose_vertex[b.prow_pose[r]]]; for (int a = 0; a < 6; ++a) xout[o + a] = xin[(size_t)6 * r + a]; }
  for (int r = 0; r < b.V.nLr; ++r) { const int o = hidx[b.lm_vertex[b.lrow_lm[r]]]; for (int a = 0; a < 3; ++a) xout[o + a] = xin[(size_t)6 * b.V.nPr + (size_t)3 * r + a]; }
}
static void from_g2o_order(const sslam_graph* h, const double* xin, std::vector<double>& xout) {
  const Batch& b = *h->batch;
  std::vector<int> hidx;
  hessian_indices(h->g, hidx);
  xout.assign((size_t)6 * b.V.nPr + (size_t)3 * b.V.nLr, 0.0);
  for (int r = 0; r < b.V.nPr; ++r) { const int o = hidx[b.pose_vertex[b.prow_pose[r]]]; for (int a = 0; a < 6; ++a) xout[(size_t)6 * r + a] = xin[o + a]; }
  for (int r = 0; r < b.V.nLr; ++r) { const int o = hidx[b.lm_vertex[b.lrow_lm[r]]]; for (int a = 0; a < 3; ++a) xout[(size_t)6 * b.V.nPr + (size_t)3 * r + a] = xin[o + a]; }
}

int sslam_graph_solve(sslam_graph* h, double lambda, double* x, int64_t* solver_iterations) {
  if (!h || !x) return set_error(SSLAM_ERR_INVALID, "null argument");
  int rc;
  if (!h->linearized && (rc = do_linearize(h))) return rc;
  Batch& b = *h->batch;
  hipLaunchKernelGGL(k_set_trial_all, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, lambda);
  if ((rc = batch_solve(b))) return rc;
  std::vector<double> xi((size_t)6 * b.V.nPr + (size_t)3 * b.V.nLr);
  LmState s;
  int fail = 0;
  SSLAM_HIP_TRY(hipMemcpyAsync(xi.data(), b.V.x, xi.size() * 8, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipMemcpyAsync(&s, b.V.lm, sizeof s, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipMemcpyAsync(&fail, b.V.pcg_fail, sizeof fail, hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  if (solver_iterations) *solver_iterations = s.pcg_iters;
  to_g2o_order(h, xi, x);
  if ((rc = chol_flow_check(b))) return rc;   // a dependency wait of the single-launch solve that gave up (also raises `fail`)
  if (fail) return set_error(SSLAM_ERR_NUMERIC, "linear solve broke down");
  return 0;
}

int sslam_graph_oplus(sslam_graph* h, const double* dx) {
  if (!h || !dx) return set_error(SSLAM_ERR_INVALID, "null argument");
  int rc = ensure_batch(h);
  if (rc) return rc;
  Batch& b = *h->batch;
  if ((rc = batch_upload_estimates(b))) return rc;
  std::vector<double> xi;
  from_g2o_order(h, dx, xi);
  if (!xi.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(b.V.x, xi.data(), xi.size() * 8, hipMemcpyHostToDevice, b.stream));
  hipLaunchKernelGGL(k_set_trial_all, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, 0.0);
  hipLaunchKernelGGL(k_oplus, dim3(vert_blocks(b)), dim3(256), 0, b.stream, b.V, b.V.x);
  if (b.V.nPose) SSLAM_HIP_TRY(hipMemcpyAsync(b.V.pose, b.V.pose_trial, (size_t)b.V.nPose * 64, hipMemcpyDeviceToDevice, b.stream));
  if (b.V.nLm) SSLAM_HIP_TRY(hipMemcpyAsync(b.V.lmk, b.V.lmk_trial, (size_t)b.V.nLm * 32, hipMemcpyDeviceToDevice, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  h->linearized = false;
  return batch_download_estimates(b);
}

// Blocks (row vertex vr, column vertex vc) of H^-1 at the current linearisation: the columns of H^-1 that belong to the
// requested column vertices are computed (Cholesky: one multi right-hand-side solve; PCG: one solve per column) and the
// requested row blocks are read out of them.  out: packed row-major d(vr) x d(vc) blocks, zeros for fixed / edge-less vertices.
static int marginal_blocks(sslam_graph* h, const std::vector<std::pair<int, int>>& pairs, double* out) {
  int rc;
  if ((rc = do_linearize(h))) return rc;  // undamped H at the current estimates
  Batch& b = *h->batch;
  std::vector<int> hidx;
  const int dim = hessian_indices(h->g, hidx);
  for (auto& pr : pairs)
    if (pr.first < 0 || pr.first >= h->g.nv() || pr.second < 0 || pr.second >= h->g.nv()) return set_error(SSLAM_ERR_INVALID, "bad vertex id (%d, %d)", pr.first, pr.second);
  // unique column vertices, their scalar columns
  std::vector<int> colv;
  for (auto& pr : pairs) if (hidx[pr.second] >= 0 && hidx[pr.first] >= 0) colv.push_back(pr.second);
  std::sort(colv.begin(), colv.end());
  colv.erase(std::unique(colv.begin(), colv.end()), colv.end());
  std::vector<int> col0(h->g.nv(), -1);   // first scalar column of a column vertex in X
  int nrhs = 0;
  for (int v : colv) { col0[v] = nrhs; nrhs += vertex_dim(h->g.vtype[v]); }
  const size_t idim = (size_t)6 * b.V.nPr + (size_t)3 * b.V.nLr;
  std::vector<double> X;   // [scalar column][g2o order]; sized when the general path is taken (11 MB at 450 keyframes: not for the path marginals)
  std::vector<double> rhs_g2o(dim, 0.0), rhs_int, xi(idim);
  if (h->g.opt.solver != 0) {
    // factor the undamped H once, then solve all unit right-hand sides together
    hipLaunchKernelGGL(k_set_trial_all, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, 0.0);
    if (!b.chol && (rc = chol_plan_build(b))) return rc;
    if ((rc = chol_plan_flow(b) ? chol_factor_flat_flow(b) : chol_factor_and_forward(b, /*flat=*/true))) return rc;
    int fail = 0;
    SSLAM_HIP_TRY(hipMemcpyAsync(&fail, b.V.pcg_fail, sizeof fail, hipMemcpyDeviceToHost, b.stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
    if ((rc = chol_flow_check(b))) return rc;
    if (fail) return set_error(SSLAM_ERR_NUMERIC, "H is not positive definite: no marginals");
    // Diagonal blocks only (what computeLandmarkMarginals asks for, semantic_graph_slam.cpp:188-190): one launch, a forward substitution
    // along each vertex' path of the elimination tree (k_chol_marginal_paths) -- no right-hand side matrix, no backward solve, and only
    // the requested blocks cross PCIe.  Off-diagonal pairs (or SSLAM_MARGINAL_PATHS=0) take the multi right-hand-side solves below.
    static const bool paths_on = [] { const char* e = getenv("SSLAM_MARGINAL_PATHS"); return !(e && atoi(e) == 0); }();
    bool all_diag = paths_on;
    for (auto& pr : pairs) all_diag = all_diag && pr.first == pr.second;
    if (all_diag) {
      std::vector<int> xoffs, dims, slot(pairs.size(), -1);
      for (size_t k = 0; k < pairs.size(); ++k) {
        const int v = pairs[k].first;
        if (hidx[v] < 0) continue;
        int xo;
        if (h->g.vtype[v] == VT_SE3) xo = 6 * b.pose_row[b.v2pose[0][v]];
        else xo = 6 * b.V.nPr + 3 * b.lm_row[b.v2lm[0][v]];
        slot[k] = (int)xoffs.size();
        xoffs.push_back(xo); dims.push_back(vertex_dim(h->g.vtype[v]));
      }
      std::vector<double> Z(xoffs.size() * 36 + 1);
      rc = chol_marginal_diag(b, xoffs, dims, Z.data());
      if (rc == 0) {
        size_t o = 0;
        for (size_t k = 0; k < pairs.size(); ++k) {
          const int d = vertex_dim(h->g.vtype[pairs[k].first]);
          for (int e = 0; e < d * d; ++e) out[o + e] = slot[k] < 0 ? 0.0 : Z[(size_t)slot[k] * 36 + e];
          o += (size_t)d * d;
        }
        return 0;
      }
      if (rc != SSLAM_ERR_UNSUPPORTED) return rc;   // a path longer than one wave's LDS holds: the general path below
    }
    X.resize((size_t)nrhs * dim);
    std::vector<double> R((size_t)nrhs * idim, 0.0), Xi((size_t)nrhs * idim);
    for (int v : colv)
      for (int c = 0; c < vertex_dim(h->g.vtype[v]); ++c) {
        rhs_g2o[hidx[v] + c] = 1.0;
        from_g2o_order(h, rhs_g2o.data(), rhs_int);
        rhs_g2o[hidx[v] + c] = 0.0;
        std::copy(rhs_int.begin(), rhs_int.end(), R.begin() + (size_t)(col0[v] + c) * idim);
      }
    if ((rc = chol_solve_multi(b, R.data(), nrhs, Xi.data()))) return rc;
    for (int q = 0; q < nrhs; ++q) {
      std::copy(Xi.begin() + (size_t)q * idim, Xi.begin() + (size_t)(q + 1) * idim, xi.begin());
      to_g2o_order(h, xi, X.data() + (size_t)q * dim);
    }
  } else {
    X.resize((size_t)nrhs * dim);
    for (int v : colv)
      for (int c = 0; c < vertex_dim(h->g.vtype[v]); ++c) {
        rhs_g2o[hidx[v] + c] = 1.0;
        from_g2o_order(h, rhs_g2o.data(), rhs_int);
        rhs_g2o[hidx[v] + c] = 0.0;
        SSLAM_HIP_TRY(hipMemcpyAsync(b.V.bvec, rhs_int.data(), rhs_int.size() * 8, hipMemcpyHostToDevice, b.stream));
        hipLaunchKernelGGL(k_set_trial_all, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, 0.0);
        if ((rc = batch_solve(b))) return rc;
        SSLAM_HIP_TRY(hipMemcpyAsync(xi.data(), b.V.x, xi.size() * 8, hipMemcpyDeviceToHost, b.stream));
        SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
        to_g2o_order(h, xi, X.data() + (size_t)(col0[v] + c) * dim);
      }
    h->linearized = false;  // b was overwritten
  }
  size_t o = 0;
  for (auto& pr : pairs) {
    const int vr = pr.first, vc = pr.second;
    const int dr = vertex_dim(h->g.vtype[vr]), dc = vertex_dim(h->g.vtype[vc]);
    for (int r = 0; r < dr; ++r)
      for (int c = 0; c < dc; ++c)
        out[o + r * dc + c] = (hidx[vr] < 0 || hidx[vc] < 0) ? 0.0 : X[(size_t)(col0[vc] + c) * dim + hidx[vr] + r];
    o += (size_t)dr * dc;
  }
  return 0;
}

int sslam_graph_marginals(sslam_graph* h, const int* ids, int n, double* out) {
  if (!h || !ids || !out || n < 0) return set_error(SSLAM_ERR_INVALID, "null argument");
  std::vector<std::pair<int, int>> pairs(n);
  for (int k = 0; k < n; ++k) pairs[k] = {ids[k], ids[k]};
  return marginal_blocks(h, pairs, out);
}

int sslam_graph_marginals_by_hessian_index(sslam_graph* h, const int* row_col, int n, double* out) {
  if (!h || !row_col || !out || n < 0) return set_error(SSLAM_ERR_INVALID, "null argument");
  std::vector<int> hidx;
  hessian_indices(h->g, hidx);
  std::unordered_map<int, int> vert_of;   // hessian index -> vertex id
  for (int v = 0; v < h->g.nv(); ++v) if (hidx[v] >= 0) vert_of[hidx[v]] = v;
  std::vector<std::pair<int, int>> pairs(n);
  for (int k = 0; k < n; ++k) {
    auto ir = vert_of.find(row_col[2 * k]), ic = vert_of.find(row_col[2 * k + 1]);
    if (ir == vert_of.end() || ic == vert_of.end())
      return set_error(SSLAM_ERR_INVALID, "(%d, %d) is not a pair of hessian indices of active vertices", row_col[2 * k], row_col[2 * k + 1]);
    pairs[k] = {ir->second, ic->second};
  }
  return marginal_blocks(h, pairs, out);
}

// ---- g2o text format (GraphSLAM::save, graph_slam.cpp:236-239; SURVEY §5 checkpoint row) ------
int sslam_graph_save_g2o(const sslam_graph* h, const char* path) {
  if (!h || !path) return set_error(SSLAM_ERR_INVALID, "null argument");
  FILE* f = fopen(path, "w");
  if (!f) return set_error(SSLAM_ERR_IO, "cannot open %s for writing", path);
  const HostGraph& G = h->g;
  fprintf(f, "PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1\n");
  for (int v = 0; v < G.nv(); ++v) {
    const double* e = &G.est[(size_t)v * 7];
    if (G.vtype[v] == VT_SE3) fprintf(f, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", v, e[0], e[1], e[2], e[3], e[4], e[5], e[6]);
    else if (G.vtype[v] == VT_POINT) fprintf(f, "VERTEX_TRACKXYZ %d %.17g %.17g %.17g\n", v, e[0], e[1], e[2]);
    else fprintf(f, "VERTEX_PLANE %d %.17g %.17g %.17g %.17g\n", v, e[0], e[1], e[2], e[3]);
    if (G.vfixed[v]) fprintf(f, "FIX %d\n", v);
  }
  for (int k = 0; k < G.ne(); ++k) {
    const double* z = &G.meas[(size_t)k * 7];
    const double* W = &G.info[(size_t)k * 36];
    if (G.etype[k] == ET_SE3) {
      fprintf(f, "EDGE_SE3:QUAT %d %d", G.evi[k], G.evj[k]);
      for (int c = 0; c < 7; ++c) fprintf(f, " %.17g", z[c]);
      for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) fprintf(f, " %.17g", W[r * 6 + c]);
    } else if (G.etype[k] == ET_SE3_POINT) {
      fprintf(f, "EDGE_SE3_TRACKXYZ %d %d 0", G.evi[k], G.evj[k]);
      for (int c = 0; c < 3; ++c) fprintf(f, " %.17g", z[c]);
      for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) fprintf(f, " %.17g", W[r * 3 + c]);
    } else if (G.etype[k] == ET_POINT_POINT) {   // g2o::EdgePointXYZ
      fprintf(f, "EDGE_POINTXYZ %d %d", G.evi[k], G.evj[k]);
      for (int c = 0; c < 3; ++c) fprintf(f, " %.17g", z[c]);
      for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) fprintf(f, " %.17g", W[r * 3 + c]);
    } else {  // row format of the in-tree EdgeSE3Plane::write (edge_se3_plane.hpp:40-47)
      fprintf(f, "EDGE_SE3_PLANE %d %d", G.evi[k], G.evj[k]);
      for (int c = 0; c < 4; ++c) fprintf(f, " %.17g", z[c]);
      for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) fprintf(f, " %.17g", W[r * 3 + c]);
    }
    fprintf(f, "\n");
  }
  fclose(f);
  return 0;
}

int sslam_graph_load_g2o(sslam_graph* h, const char* path) {
  if (!h || !path) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (h->g.nv() != 0) return set_error(SSLAM_ERR_INVALID, "load requires an empty graph");
  FILE* f = fopen(path, "r");
  if (!f) return set_error(SSLAM_ERR_IO, "cannot open %s", path);
  char tag[64];
  std::unordered_map<int, int> idmap;  // file id -> vertex id (ids must be dense & ordered for exact round trips)
  auto rd = [&](double* a, int n) { for (int k = 0; k < n; ++k) if (fscanf(f, "%lf", a + k) != 1) return false; return true; };
  int rc = 0;
  while (fscanf(f, "%63s", tag) == 1) {
    const std::string t(tag);
    if (t == "VERTEX_SE3:QUAT") { int id; double e[7]; if (fscanf(f, "%d", &id) != 1 || !rd(e, 7)) { rc = -1; break; } idmap[id] = add_vertex(h, VT_SE3, e, 7, 0); }
    else if (t == "VERTEX_TRACKXYZ") { int id; double e[3]; if (fscanf(f, "%d", &id) != 1 || !rd(e, 3)) { rc = -1; break; } idmap[id] = add_vertex(h, VT_POINT, e, 3, 0); }
    else if (t == "VERTEX_PLANE") { int id; double e[4]; if (fscanf(f, "%d", &id) != 1 || !rd(e, 4)) { rc = -1; break; } idmap[id] = add_vertex(h, VT_PLANE, e, 4, 0); }
    else if (t == "FIX") { int id; if (fscanf(f, "%d", &id) != 1 || !idmap.count(id)) { rc = -1; break; } h->g.vfixed[idmap[id]] = 1; }
    else if (t == "EDGE_SE3:QUAT") {
      int i, j; double z[7], u[21], W[36];
      if (fscanf(f, "%d %d", &i, &j) != 2 || !rd(z, 7) || !rd(u, 21) || !idmap.count(i) || !idmap.count(j)) { rc = -1; break; }
      int q = 0; for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { W[r * 6 + c] = u[q]; W[c * 6 + r] = u[q]; ++q; }
      if (add_edge(h, ET_SE3, idmap[i], idmap[j], z, 7, W, 6) < 0) { rc = -1; break; }
    } else if (t == "EDGE_SE3_TRACKXYZ") {
      int i, j, pid; double z[3], u[6], W[9];
      if (fscanf(f, "%d %d %d", &i, &j, &pid) != 3 || !rd(z, 3) || !rd(u, 6) || !idmap.count(i) || !idmap.count(j)) { rc = -1; break; }
      int q = 0; for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) { W[r * 3 + c] = u[q]; W[c * 3 + r] = u[q]; ++q; }
      if (add_edge(h, ET_SE3_POINT, idmap[i], idmap[j], z, 3, W, 3) < 0) { rc = -1; break; }
    } else if (t == "EDGE_POINTXYZ") {
      int i, j; double z[3], u[6], W[9];
      if (fscanf(f, "%d %d", &i, &j) != 2 || !rd(z, 3) || !rd(u, 6) || !idmap.count(i) || !idmap.count(j)) { rc = -1; break; }
      int q = 0; for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) { W[r * 3 + c] = u[q]; W[c * 3 + r] = u[q]; ++q; }
      if (add_edge(h, ET_POINT_POINT, idmap[i], idmap[j], z, 3, W, 3) < 0) { rc = -1; break; }
    } else if (t == "EDGE_SE3_PLANE") {
      int i, j; double z[4], u[6], W[9];
      if (fscanf(f, "%d %d", &i, &j) != 2 || !rd(z, 4) || !rd(u, 6) || !idmap.count(i) || !idmap.count(j)) { rc = -1; break; }
      int q = 0; for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) { W[r * 3 + c] = u[q]; W[c * 3 + r] = u[q]; ++q; }
      if (add_edge(h, ET_SE3_PLANE, idmap[i], idmap[j], z, 4, W, 3) < 0) { rc = -1; break; }
    } else {  // PARAMS_SE3OFFSET and unknown rows: skip to end of line
      int ch; while ((ch = fgetc(f)) != EOF && ch != '\n') {}
    }
  }
  fclose(f);
  if (rc) return set_error(SSLAM_ERR_IO, "parse error in %s near tag %s", path, tag);
  return 0;
}

// ---- batch API --------------------------------------------------------------------------------
static sslam_batch* batch_create_single(sslam_graph* const* graphs, int n);
sslam_batch* sslam_batch_create(sslam_graph* const* graphs, int n) {
  const char* e = getenv("SSLAM_BATCH_STREAMS");
  return sslam_batch_create_streams(graphs, n, e ? atoi(e) : 1);
}
sslam_batch* sslam_batch_create_streams(sslam_graph* const* graphs, int n, int n_streams) {
  if (!graphs || n <= 0) { set_error(SSLAM_ERR_INVALID, "empty batch"); return nullptr; }
  const int K = std::min(std::max(n_streams, 1), std::min(n, 64));
  if (K == 1) return batch_create_single(graphs, n);
  for (int i = 0; i < n; ++i) {   // the same rule as inside one batch, across the parts
    if (!graphs[i]) { set_error(SSLAM_ERR_INVALID, "null graph"); return nullptr; }
    const Options &oa = graphs[0]->g.opt, &ob = graphs[i]->g.opt;
    if (graphs[i]->g.device != graphs[0]->g.device || oa.solver != ob.solver || oa.pcg_tol != ob.pcg_tol || oa.pcg_max_iters != ob.pcg_max_iters || oa.dcs_phi != ob.dcs_phi) {
      set_error(SSLAM_ERR_INVALID, "batch graphs must share one device and their solver options (graph %d differs from graph 0)", i); return nullptr;
    }
  }
  sslam_batch* h = new sslam_batch();
  h->b.device = graphs[0]->g.device;
  for (int k = 0; k < K; ++k) {
    const int lo = (int)((int64_t)n * k / K), hi = (int)((int64_t)n * (k + 1) / K);
    sslam_batch* part = batch_create_single(graphs + lo, hi - lo);
    if (!part) { delete h; return nullptr; }
    h->parts.push_back(part);
    h->part0.push_back(lo);
  }
  h->part0.push_back(n);
  return h;
}
static sslam_batch* batch_create_single(sslam_graph* const* graphs, int n) {
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback"); return nullptr; }
  sslam_batch* h = new sslam_batch();
  h->b.device = graphs[0]->g.device;
  for (int i = 0; i < n; ++i) {
    if (!graphs[i] || graphs[i]->g.device != h->b.device) { set_error(SSLAM_ERR_INVALID, "batch graphs must share one device"); delete h; return nullptr; }
    const Options &oa = graphs[0]->g.opt, &ob = graphs[i]->g.opt;   // one set of solver options drives the whole batch
    if (oa.solver != ob.solver || oa.pcg_tol != ob.pcg_tol || oa.pcg_max_iters != ob.pcg_max_iters || oa.dcs_phi != ob.dcs_phi) {
      set_error(SSLAM_ERR_INVALID, "batch graphs must share their solver options (graph %d differs from graph 0)", i); delete h; return nullptr;
    }
    h->b.graphs.push_back(&graphs[i]->g);
  }
  if (batch_build(h->b) != 0) { delete h; return nullptr; }
  return h;
}
void sslam_batch_destroy(sslam_batch* h) { delete h; }
// A batch is compiled for the structure its graphs had at sslam_batch_create; the graphs must outlive the batch and must
// not gain vertices / edges afterwards (estimates may change: sslam_graph_set_vertex + sslam_batch_upload).
static int batch_check(sslam_batch* h) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null batch");
  for (sslam_batch* p : h->parts) { const int rc = batch_check(p); if (rc) return rc; }
  const Batch& b = h->b;
  for (size_t g = 0; g < b.graphs.size(); ++g)
    if (b.versions[g] != b.graphs[g]->structure_version)
      return set_error(SSLAM_ERR_INVALID, "graph %zu of the batch gained vertices or edges after sslam_batch_create: create a new batch", g);
  return 0;
}
int sslam_batch_upload(sslam_batch* h) {
  const int rc = batch_check(h);
  if (rc) return rc;
  if (!h->parts.empty()) return for_each_part(h, false, [](sslam_batch* p, int) { return batch_upload_estimates(p->b); });
  return batch_upload_estimates(h->b);
}
int sslam_batch_download(sslam_batch* h) {
  const int rc = batch_check(h);
  if (rc) return rc;
  if (!h->parts.empty()) return for_each_part(h, false, [](sslam_batch* p, int) { return batch_download_estimates(p->b); });
  return batch_download_estimates(h->b);
}
int sslam_batch_optimize(sslam_batch* h, int max_iters, sslam_opt_stats* out) {
  if (!h || !out) return set_error(SSLAM_ERR_INVALID, "null argument");
  const int rc = batch_check(h);
  if (rc) return rc;
  if (!h->parts.empty())
    return for_each_part(h, true, [&](sslam_batch* p, int k) { return batch_optimize(p->b, max_iters, out + h->part0[k]); });
  return batch_optimize(h->b, max_iters, out);
}
// ---- edge-sharded mode (SURVEY 8e mode E; BASELINE.json configs[4]): the edges of every graph of the batch are split
//      contiguously over the ranks, each rank builds the partial normal equations of its edges, ONE RCCL all-reduce of the
//      contiguous [H || b] buffer sums them, and the rest of the LM step runs replicated.
static int batch_set_shard(Batch& b, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return set_error(SSLAM_ERR_INVALID, "bad rank %d of %d", rank, world);
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  std::vector<int> lo(b.graphs.size()), hi(b.graphs.size());
  for (size_t g = 0; g < b.graphs.size(); ++g) {
    const int n = b.graphs[g]->ne();
    const int base = n / world, extra = n % world;
    lo[g] = rank * base + std::min(rank, extra);
    hi[g] = lo[g] + base + (rank < extra ? 1 : 0);
    if (world == 1) { lo[g] = 0; hi[g] = 0x7fffffff; }
  }
  SSLAM_HIP_TRY(hipMemcpyAsync((void*)b.V.shard_lo, lo.data(), lo.size() * sizeof(int), hipMemcpyHostToDevice, b.stream));
  SSLAM_HIP_TRY(hipMemcpyAsync((void*)b.V.shard_hi, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  b.sharded = world > 1 || b.comm != nullptr;   // a forced single-rank communicator keeps the whole sharded path (masked kernels + all-reduce) on
  b.shard_rank = rank; b.shard_world = world;
  if (b.sharded && !b.d_hb_part) {
    int rc;
    if ((rc = dev_alloc(b, (size_t)b.hb_doubles, &b.d_hb_part))) return rc;
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  }
  return 0;
}
int sslam_comm_unique_id(char id_out[128]) {
  if (!id_out) return set_error(SSLAM_ERR_INVALID, "null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!rccl_api().ok()) return set_error(SSLAM_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", dlerror());
  ncclUniqueId id;
  const ncclResult_t r = rccl_api().GetUniqueId(&id);
  if (r != ncclSuccess) return set_error(SSLAM_ERR_HIP, "ncclGetUniqueId: %s", rccl_api().GetErrorString(r));
  memcpy(id_out, &id, sizeof id);
  return 0;
}
int sslam_batch_comm_init(sslam_batch* h, const char id_in[128], int rank, int world) {
  if (!h || !id_in) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (!h->parts.empty()) return set_error(SSLAM_ERR_UNSUPPORTED, "not available on a stream group: the edge-sharded mode and the [H || b] read-back work on single-stream batches");
  Batch& b = h->b;
  int rc = batch_check(h);
  if (rc) return rc;
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  if (b.comm) { batch_comm_destroy(b.comm); b.comm = nullptr; }
  // world == 1 switches the mode off -- unless SSLAM_FORCE_COMM=1 asks for a single-rank communicator: the whole product path of the
  // mode (shard-masked kernels, ncclCommInitRank, the out-of-place ncclAllReduce on the batch's stream) then runs on one GPU
  const char* force = getenv("SSLAM_FORCE_COMM");
  if (world > 1 || (force && atoi(force) != 0)) {
    if (!rccl_api().ok()) return set_error(SSLAM_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", dlerror());
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl_api().CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return set_error(SSLAM_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_api().GetErrorString(r));
    b.comm = comm;
  }
  return batch_set_shard(b, rank, world);
}
int sslam_batch_set_edge_shard(sslam_batch* h, int rank, int world) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null batch");
  if (!h->parts.empty()) return set_error(SSLAM_ERR_UNSUPPORTED, "not available on a stream group: the edge-sharded mode and the [H || b] read-back work on single-stream batches");
  int rc = batch_check(h);
  if (rc) return rc;
  return batch_set_shard(h->b, rank, world);
}
int64_t sslam_batch_linearize_hb(sslam_batch* h, double* h_and_b, int64_t capacity) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null batch");
  if (!h->parts.empty()) return set_error(SSLAM_ERR_UNSUPPORTED, "not available on a stream group: the edge-sharded mode and the [H || b] read-back work on single-stream batches");
  Batch& b = h->b;
  if (!h_and_b) return b.hb_doubles;
  if (capacity < b.hb_doubles) return set_error(SSLAM_ERR_INVALID, "buffer of %lld doubles needed", (long long)b.hb_doubles);
  int rc = batch_check(h);
  if (rc) return rc;
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  if (!b.uploaded && (rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, b.V.pose, b.V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(b.V.B), dim3(64), 0, b.stream, b.V, b.d_part_e, 0);
  if ((rc = batch_linearize(b))) return rc;
  SSLAM_HIP_TRY(hipMemcpyAsync(h_and_b, b.V.Hpp_diag, (size_t)b.hb_doubles * sizeof(double), hipMemcpyDeviceToHost, b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  b.harvest();
  return b.hb_doubles;
}

int sslam_batch_time_linearize(sslam_batch* h, int repeats, double* ms_per_build) {
  if (!h || !ms_per_build || repeats <= 0) return set_error(SSLAM_ERR_INVALID, "bad argument");
  if (!h->parts.empty()) {   // the parts one after the other: their times add up to one build of every graph
    *ms_per_build = 0;
    return for_each_part(h, false, [&](sslam_batch* p, int) { double ms = 0; const int rc = sslam_batch_time_linearize(p, repeats, &ms); *ms_per_build += ms; return rc; });
  }
  Batch& b = h->b;
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  int rc;
  if (!b.uploaded && (rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, b.V.pose, b.V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(b.V.B), dim3(64), 0, b.stream, b.V, b.d_part_e, 0);
  const bool prof = b.profiling;
  b.profiling = false;
  if ((rc = batch_linearize(b))) return rc;  // warm-up
  hipEvent_t e0, e1;
  SSLAM_HIP_TRY(hipEventCreate(&e0)); SSLAM_HIP_TRY(hipEventCreate(&e1));
  SSLAM_HIP_TRY(hipEventRecord(e0, b.stream));
  for (int k = 0; k < repeats; ++k) if ((rc = batch_linearize(b))) return rc;
  SSLAM_HIP_TRY(hipEventRecord(e1, b.stream));
  SSLAM_HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  SSLAM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  b.profiling = prof;
  *ms_per_build = (double)ms / repeats;
  return 0;
}
// one numeric factorisation (+ fused forward solve) and one backward solve of EVERY graph of the batch, `repeats` times: mean
// milliseconds by hipEvents on the batch's stream (the LM loop's own launches mix full and partial rounds)
int sslam_batch_time_solver(sslam_batch* h, int repeats, double* factor_ms, double* solve_ms) {
  if (!h || repeats <= 0 || !factor_ms || !solve_ms) return set_error(SSLAM_ERR_INVALID, "bad argument");
  if (!h->parts.empty()) {
    *factor_ms = *solve_ms = 0;
    return for_each_part(h, false, [&](sslam_batch* p, int) {
      double f = 0, v = 0; const int rc = sslam_batch_time_solver(p, repeats, &f, &v); *factor_ms += f; *solve_ms += v; return rc; });
  }
  Batch& b = h->b;
  if (b.graphs[0]->opt.solver == 0 || b.graphs[0]->opt.solver == 2) return set_error(SSLAM_ERR_UNSUPPORTED, "direct solvers only");
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  int rc;
  if ((rc = chol_set_active(b, nullptr))) return rc;   // a stale compaction (an optimise that failed half way) must not shrink what is timed
  if (!b.uploaded && (rc = batch_upload_estimates(b))) return rc;
  if ((rc = batch_chi2(b, b.V.pose, b.V.lmk, 0))) return rc;
  hipLaunchKernelGGL(k_lm_init, dim3(b.V.B), dim3(64), 0, b.stream, b.V, b.d_part_e, 0);
  if ((rc = batch_linearize(b))) return rc;
  hipLaunchKernelGGL(k_set_trial_all, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, 1.0);
  struct Restore {   // events and the profiling switch are put back on every return path
    Batch& b; bool prof; hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    ~Restore() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); b.profiling = prof; }
  } guard{b, b.profiling};
  b.profiling = false;
  if ((rc = batch_solve(b))) return rc;   // warm-up (builds the plan)
  SSLAM_HIP_TRY(hipEventCreate(&guard.e[0])); SSLAM_HIP_TRY(hipEventCreate(&guard.e[1])); SSLAM_HIP_TRY(hipEventCreate(&guard.e[2]));
  hipEvent_t e0 = guard.e[0], e1 = guard.e[1], e2 = guard.e[2];
  double tf = 0, ts = 0;
  for (int k = 0; k < repeats; ++k) {
    SSLAM_HIP_TRY(hipEventRecord(e0, b.stream));
    if ((rc = chol_factor_and_forward(b))) return rc;
    SSLAM_HIP_TRY(hipEventRecord(e1, b.stream));
    if ((rc = chol_backward(b))) return rc;
    SSLAM_HIP_TRY(hipEventRecord(e2, b.stream));
    SSLAM_HIP_TRY(hipEventSynchronize(e2));
    float a = 0, c = 0;
    SSLAM_HIP_TRY(hipEventElapsedTime(&a, e0, e1)); SSLAM_HIP_TRY(hipEventElapsedTime(&c, e1, e2));
    tf += a; ts += c;
  }
  *factor_ms = tf / repeats; *solve_ms = ts / repeats;
  return 0;
}
int64_t sslam_batch_linearize_bytes(const sslam_batch* h) {
  if (!h) return 0;
  if (!h->parts.empty()) { int64_t t = 0; for (const sslam_batch* p : h->parts) t += sslam_batch_linearize_bytes(p); return t; }
  const Batch& b = h->b;
  // 344*Eo + 160*El(176 plane) + 288*Np + 72*Nl + 288*Eo + 144*El + 48*Np + 24*Nl
  int64_t bytes = 0;
  for (size_t g = 0; g < b.graphs.size(); ++g) {
    const HostGraph& G = *b.graphs[g];
    for (int k = 0; k < G.ne(); ++k)
      bytes += G.etype[k] == ET_SE3 ? 344 + 288 : (G.etype[k] == ET_SE3_POINT ? 160 + 144 : (G.etype[k] == ET_SE3_PLANE ? 176 + 144 : 8 + 24 + 48 + 48 + 72));
  }
  bytes += (int64_t)b.V.nPr * (288 + 48) + (int64_t)b.V.nLr * (72 + 24);
  return bytes;
}
int sslam_batch_info(sslam_batch* h, const char* key, double* value) {
  if (!h || !key || !value) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (!h->parts.empty()) {   // sums over the parts; the depth of the plans is their maximum
    const std::string kk(key);
    if (kk == "streams") { *value = (double)h->parts.size(); return 0; }
    double acc = 0;
    for (sslam_batch* p : h->parts) {
      double v = 0;
      const int rc = sslam_batch_info(p, key, &v);
      if (rc) return rc;
      acc = kk == "factor_levels" ? std::max(acc, v) : acc + v;
    }
    *value = acc;
    return 0;
  }
  if (std::string(key) == "streams") { *value = 1; return 0; }
  Batch& b = h->b;
  const std::string k(key);
  int rc;
  if ((k == "factor_lnz" || k == "factor_levels" || k == "factor_launches" || k == "factor_bytes") && !b.chol && (rc = chol_plan_build(b))) return rc;
  const double dim = 6.0 * b.V.nPr + 3.0 * b.V.nLr;
  if (k == "factor_lnz") *value = (double)chol_plan_lnz(b);
  else if (k == "factor_levels") *value = (double)chol_plan_levels(b);
  else if (k == "factor_launches") *value = (double)chol_plan_launches(b);
  else if (k == "h_doubles") *value = (double)b.V.h_total;
  else if (k == "dim") *value = dim;
  else if (k == "allreduce_calls") *value = (double)b.allreduce_calls;
  else if (k == "factor_bytes") *value = 8.0 * ((double)b.V.h_total + dim + (double)chol_plan_lnz(b) + dim);
  else return set_error(SSLAM_ERR_INVALID, "unknown info key '%s'", key);
  return 0;
}
int sslam_batch_set_profiling(sslam_batch* h, int enable) {
  if (!h) return set_error(SSLAM_ERR_INVALID, "null batch");
  for (sslam_batch* p : h->parts) sslam_batch_set_profiling(p, enable);
  h->b.profiling = enable != 0;
  h->b.timers.clear();
  return 0;
}
int sslam_batch_kernel_time(sslam_batch* h, const char* name, double* total_ms, int64_t* launches) {
  if (!h || !name) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (!h->parts.empty()) {   // per-stream event times, summed: the parts' kernels overlap on the chip, so the sum exceeds the wall time
    double t = 0; int64_t n = 0;
    for (sslam_batch* p : h->parts) { double a = 0; int64_t c = 0; sslam_batch_kernel_time(p, name, &a, &c); t += a; n += c; }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    return 0;
  }
  auto it = h->b.timers.find(name);
  if (total_ms) *total_ms = it == h->b.timers.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->b.timers.end() ? 0 : it->second.launches;
  return 0;
}

// ---- plan introspection (host only, no device needed): the symbolic Cholesky plan of a batch as flat int32 arrays ----------
struct sslam_debug_plan { CholHost H; int64_t h_total = 0; std::vector<int> ppoff, plblk; int sc[16] = {0}; };
void* sslam_debug_plan_create(sslam_graph* const* graphs, int n) {
  if (!graphs || n <= 0) { set_error(SSLAM_ERR_INVALID, "empty batch"); return nullptr; }
  Batch b;
  for (int i = 0; i < n; ++i) { if (!graphs[i]) { set_error(SSLAM_ERR_INVALID, "null graph"); return nullptr; } b.graphs.push_back(&graphs[i]->g); }
  static const bool timing = getenv("SSLAM_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (batch_build(b, true) != 0) return nullptr;
  const auto t1 = std::chrono::steady_clock::now();
  SymIn in;
  chol_sym_input(b, in);
  CholOpts opt;
  opt.from_env();
  chol_opts_normalise(opt, b.V.B, b.V.nPr + b.V.nLr);   // the plan chol_plan_build would execute, small-batch regime included
  sslam_debug_plan* P = new sslam_debug_plan();
  P->h_total = b.V.h_total;
  for (auto& q : b.ppoff) { P->ppoff.push_back(q.first); P->ppoff.push_back(q.second); }
  for (auto& q : b.plblk) { P->plblk.push_back(q.first); P->plblk.push_back(q.second); }
  P->sc[11] = b.V.nPr; P->sc[12] = b.V.nLr; P->sc[13] = (int)b.hll_base; P->sc[14] = (int)b.hpp_off_base; P->sc[15] = (int)b.hpl_base;
  if (chol_symbolic(in, opt, P->H)) { set_error(SSLAM_ERR_NUMERIC, "Cholesky plan: %s", P->H.error.c_str()); delete P; return nullptr; }
  if (timing) fprintf(stderr, "[timing] host plan: batch tables %.3f ms, symbolic Cholesky %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return P;
}
void sslam_debug_plan_destroy(void* p) { delete (sslam_debug_plan*)p; }
int64_t sslam_debug_plan_array(void* p, const char* name, void* out, int64_t cap_bytes) {
  if (!p || !name) return set_error(SSLAM_ERR_INVALID, "null argument");
  const CholHost& H = ((sslam_debug_plan*)p)->H;
  const std::string k(name);
  const void* src = nullptr; int64_t bytes = 0;
  const sslam_debug_plan& DP = *(sslam_debug_plan*)p;
  int scal[17] = {H.ncol, H.nlevels, H.dim, H.B, H.npiece, (int)H.lnz, H.tail_lds_f, H.tail_lds_b, H.nt_leaf, H.nt_tail, (int)DP.h_total,
                  DP.sc[11], DP.sc[12], DP.sc[13], DP.sc[14], DP.sc[15], (int)H.unz};
#define ARR(nm, vec) if (k == nm) { src = (vec).data(); bytes = (int64_t)(vec).size() * sizeof((vec)[0]); }
  ARR("col", H.col) ARR("blk", H.blk) ARR("upd", H.upd) ARR("item", H.item) ARR("mb", H.mb) ARR("ilv", H.ilv) ARR("piece", H.piece)
  ARR("lvl_ptr", H.lvl_ptr) ARR("lvl_cols", H.lvl_cols) ARR("plv_ptr", H.plv_ptr) ARR("plv_pieces", H.plv_pieces)
  ARR("ppoff", DP.ppoff) ARR("plblk", DP.plblk) ARR("tail_ptr", H.tail_ptr)
  ARR("asrc", H.asrc) ARR("usrc", H.usrc) ARR("fwd", H.fwd) ARR("uitem", H.uitem) ARR("umb", H.umb) ARR("tail_pieces", H.tail_pieces) ARR("plv_lds_f", H.plv_lds_f) ARR("plv_lds_b", H.plv_lds_b)
  ARR("rcol", H.rcol) ARR("rupd", H.rupd) ARR("plv_nt", H.plv_nt) ARR("plv_cls", H.plv_cls)
  ARR("fblob", H.fblob) ARR("fgrp", H.fgrp) ARR("plv_lds_ff", H.plv_lds_ff)
#undef ARR
  int fscal[4] = {H.front ? 1 : 0, (int)H.funz, H.tail_lds_ff, 0};   // front tables (front_plan.hpp): present, doubles of update matrices, LDS of the tail
  if (k == "fscalars") { src = fscal; bytes = sizeof fscal; }
  if (k == "scalars") { src = scal; bytes = sizeof scal; }
  static const char* known[] = {"col", "blk", "upd", "item", "mb", "ilv", "piece", "lvl_ptr", "lvl_cols", "plv_ptr", "plv_pieces", "ppoff", "plblk",
                                "tail_ptr", "tail_pieces", "plv_lds_f", "plv_lds_b", "scalars", "asrc", "usrc", "fwd", "uitem", "umb", "rcol", "rupd", "plv_nt", "plv_cls",
                                "fblob", "fgrp", "plv_lds_ff", "fscalars"};
  bool ok = false;
  for (const char* q : known) ok |= (k == q);
  if (!ok) return set_error(SSLAM_ERR_INVALID, "unknown plan array '%s'", name);
  if (out && cap_bytes >= bytes && bytes > 0) memcpy(out, src, (size_t)bytes);
  return bytes;
}

}  // extern "C"
