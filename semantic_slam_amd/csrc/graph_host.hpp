// Host-side graph container + batch compiler for the MI355X ps_graph_slam backend.
// Mirrors what g2o::SparseOptimizer holds for the reference's GraphSLAM
// (reference include/ps_graph_slam/graph_slam.hpp:147) as plain arrays.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sslam {

enum { VT_SE3 = 0, VT_POINT = 1, VT_PLANE = 2 };
enum { ET_SE3 = 0, ET_SE3_POINT = 1, ET_SE3_PLANE = 2, ET_POINT_POINT = 3 };   // 3: g2o::EdgePointXYZ (graph_slam.cpp:168-180)

struct Options {
  int solver = 1;            // 0 PCG, 1 sparse block Cholesky
  double pcg_tol = 1e-10;    // relative residual ||r|| / ||b||
  int pcg_max_iters = 20000;
  double dcs_phi = 0.0;      // > 0: g2o::RobustKernelDCS(delta = phi) on the landmark edges (quirk B1: off by default)
  int speculative = 1;       // 1 / 2: a single small graph runs the damping trials of an LM iteration side by side in one launch (1, the default since
                             // round 5: once a trial has been rejected and while the lanes fit the chip; 2: always); same results, bitwise; 0: off
  int fused = 1;             // small batches: factor + both solves + the halves of a damping trial in one dependency-driven launch (k_chol_flow); 0: the stand-alone kernels (same results, bitwise)
};

struct HostGraph {
  int device = 0;
  std::vector<int> vtype, vfixed;
  std::vector<double> est;  // 7 per vertex
  std::vector<int> etype, evi, evj;
  std::vector<double> meas;  // 7 per edge
  std::vector<double> info;  // 36 per edge (leading d*d used, row-major)
  Options opt;
  uint64_t structure_version = 0;  // bumped on every add_*
  int nv() const { return (int)vtype.size(); }
  int ne() const { return (int)etype.size(); }
};

inline int vertex_dim(int t) { return t == VT_SE3 ? 6 : 3; }
inline int vertex_est_len(int t) { return t == VT_SE3 ? 7 : (t == VT_POINT ? 3 : 4); }

// g2o initializeOptimization ordering: non-fixed vertices that own >= 1 edge get
// consecutive scalar offsets by id.  Returns the total dimension.
inline int hessian_indices(const HostGraph& g, std::vector<int>& hidx) {
  const int nv = g.nv();
  std::vector<char> has(nv, 0);
  for (int k = 0; k < g.ne(); ++k) { has[g.evi[k]] = 1; has[g.evj[k]] = 1; }
  hidx.assign(nv, -1);
  int off = 0;
  for (int v = 0; v < nv; ++v)
    if (!g.vfixed[v] && has[v]) { hidx[v] = off; off += vertex_dim(g.vtype[v]); }
  return off;
}

}  // namespace sslam
