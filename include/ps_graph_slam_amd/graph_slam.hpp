// Drop-in C++ shim: ps_graph_slam::GraphSLAM over the MI355X C-ABI (include/sslam.h).
//
// Same class name, method names, argument meaning and return conventions as the reference class
// (reference include/ps_graph_slam/graph_slam.hpp:35-152, src/ps_graph_slam/graph_slam.cpp), with g2o's
// vertex/edge pointer types replaced by light handles exposing what the reference's callers use:
//   node->estimate()            (semantic_graph_slam.cpp:94-95, data_association.h:378)
//   node->hessianIndex()        (semantic_graph_slam.cpp:188-190)
//   node->unlockQuadraticForm() (semantic_graph_slam.cpp:187)     -- no-op here
//   node->id()
// Like g2o, the GraphSLAM object owns its vertices: add_*_node return raw pointers that stay valid for the life of the graph,
// so `keyframe->node` / `landmark.node` (keyframe.hpp:47, landmark.h:31) keep their pointer types.
// Eigen is not required: poses are sslam::Isometry (3x4 row-major R|t), points std::array<double,3>;
// when the including translation unit has Eigen available define SSLAM_WITH_EIGEN before including
// this header to get overloads taking Eigen::Isometry3d / Eigen::Vector3d / Eigen::MatrixXd exactly as
// the reference signatures do.
#ifndef PS_GRAPH_SLAM_AMD_GRAPH_SLAM_HPP
#define PS_GRAPH_SLAM_AMD_GRAPH_SLAM_HPP

#include <array>
#include <chrono>
#include <cmath>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../sslam.h"

namespace sslam {

struct Isometry {  // R (row-major 3x3) | t
  double R[9];
  double t[3];
  static Isometry Identity() { return Isometry{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
};

inline void isometry_to_tq(const Isometry& T, double tq[7]) {
  const double* R = T.R;
  double q[4];  // x y z w
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
  else { const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
  if (q[3] < 0) for (double& v : q) v = -v;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  tq[0] = T.t[0]; tq[1] = T.t[1]; tq[2] = T.t[2];
  tq[3] = q[0] / n; tq[4] = q[1] / n; tq[5] = q[2] / n; tq[6] = q[3] / n;
}
inline Isometry tq_to_isometry(const double tq[7]) {
  const double x = tq[3], y = tq[4], z = tq[5], w = tq[6];
  Isometry T;
  T.R[0] = 1 - 2 * (y * y + z * z); T.R[1] = 2 * (x * y - z * w); T.R[2] = 2 * (x * z + y * w);
  T.R[3] = 2 * (x * y + z * w); T.R[4] = 1 - 2 * (x * x + z * z); T.R[5] = 2 * (y * z - x * w);
  T.R[6] = 2 * (x * z - y * w); T.R[7] = 2 * (y * z + x * w); T.R[8] = 1 - 2 * (x * x + y * y);
  T.t[0] = tq[0]; T.t[1] = tq[1]; T.t[2] = tq[2];
  return T;
}

// what the reference reads through g2o::VertexSE3* / g2o::VertexPointXYZ*
class VertexHandle {
 public:
  VertexHandle(sslam_graph* g, int id) : g_(g), id_(id) {}
  int id() const { return id_; }
  int hessianIndex() const { return sslam_graph_hessian_index(g_, id_); }
  void unlockQuadraticForm() const {}
  virtual int dimension() const { return 3; }
  virtual ~VertexHandle() {}
 protected:
  sslam_graph* g_;
  int id_;
};
class VertexSE3 : public VertexHandle {
 public:
  using VertexHandle::VertexHandle;
  int dimension() const override { return 6; }
  Isometry estimate() const { double tq[7]; sslam_graph_get_vertex(g_, id_, tq); return tq_to_isometry(tq); }
  void setEstimate(const Isometry& T) { double tq[7]; isometry_to_tq(T, tq); sslam_graph_set_vertex(g_, id_, tq); }
};
class VertexPointXYZ : public VertexHandle {
 public:
  using VertexHandle::VertexHandle;
  std::array<double, 3> estimate() const { double p[7]; sslam_graph_get_vertex(g_, id_, p); return {p[0], p[1], p[2]}; }
};
class VertexPlane : public VertexHandle {
 public:
  using VertexHandle::VertexHandle;
  std::array<double, 4> estimate() const { double p[7]; sslam_graph_get_vertex(g_, id_, p); return {p[0], p[1], p[2], p[3]}; }
};
struct EdgeHandle { int id; };

// What the caller reads out of g2o::SparseBlockMatrix<Eigen::MatrixXd> after computeMarginals
// (semantic_graph_slam.cpp:196-203: `spinv.block(r, c)->eval().cast<float>()`): blocks of H^-1 addressed by hessian indices.
struct MarginalBlock {
  int rows = 0, cols = 0;
  std::vector<double> v;   // row-major rows x cols
  double operator()(int r, int c) const { return v[(size_t)r * cols + c]; }
#ifdef SSLAM_WITH_EIGEN
  Eigen::MatrixXd eval() const { Eigen::MatrixXd M(rows, cols); for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) M(r, c) = (*this)(r, c); return M; }
#else
  const MarginalBlock& eval() const { return *this; }
#endif
};
class SparseBlockMatrix {
 public:
  const MarginalBlock* block(int r, int c) const { auto it = blocks_.find({r, c}); return it == blocks_.end() ? nullptr : &it->second; }
  MarginalBlock* block(int r, int c, bool alloc) { if (alloc) return &blocks_[{r, c}]; auto it = blocks_.find({r, c}); return it == blocks_.end() ? nullptr : &it->second; }
  void clear() { blocks_.clear(); }
  size_t size() const { return blocks_.size(); }
 private:
  std::map<std::pair<int, int>, MarginalBlock> blocks_;
};

}  // namespace sslam

namespace ps_graph_slam {

class GraphSLAM {
 public:
  explicit GraphSLAM(bool verbose, int device = 0) : verbose_(verbose), graph(sslam_graph_create(device), &sslam_graph_destroy) {
    std::cout << "construct solver... " << std::endl;   // graph_slam.cpp:45
    if (!graph) { std::cerr << "error : failed to allocate solver!!" << std::endl; return; }
    std::cout << "done" << std::endl;
  }

  /** add_se3_node (graph_slam.cpp:104-115): the first vertex of the graph is fixed */
  sslam::VertexSE3* add_se3_node(const sslam::Isometry& pose) {
    double tq[7]; sslam::isometry_to_tq(pose, tq);
    const int id = check(sslam_graph_add_vertex_se3(graph.get(), tq, -1));
    return own(new sslam::VertexSE3(graph.get(), id));
  }
  /** add_plane_node (graph_slam.cpp:117-125, commented out upstream) */
  sslam::VertexPlane* add_plane_node(const std::array<double, 4>& plane_coeffs) {
    return own(new sslam::VertexPlane(graph.get(), check(sslam_graph_add_vertex_plane(graph.get(), plane_coeffs.data()))));
  }
  /** add_point_xyz_node (graph_slam.cpp:127-134) */
  sslam::VertexPointXYZ* add_point_xyz_node(const std::array<double, 3>& xyz) {
    return own(new sslam::VertexPointXYZ(graph.get(), check(sslam_graph_add_vertex_point(graph.get(), xyz.data()))));
  }
  /** add_se3_edge (graph_slam.cpp:136-148); information_matrix: 36 doubles row-major 6x6 */
  sslam::EdgeHandle add_se3_edge(const sslam::VertexSE3* v1, const sslam::VertexSE3* v2, const sslam::Isometry& relative_pose,
                                 const double* information_matrix) {
    double tq[7]; sslam::isometry_to_tq(relative_pose, tq);
    return {check(sslam_graph_add_edge_se3(graph.get(), v1->id(), v2->id(), tq, information_matrix))};
  }
  /** add_se3_point_xyz_edge (graph_slam.cpp:150-166); information_matrix: 9 doubles row-major 3x3 */
  sslam::EdgeHandle add_se3_point_xyz_edge(const sslam::VertexSE3* v_se3, const sslam::VertexPointXYZ* v_xyz,
                                           const std::array<double, 3>& xyz, const double* information_matrix) {
    return {check(sslam_graph_add_edge_se3_point(graph.get(), v_se3->id(), v_xyz->id(), xyz.data(), information_matrix))};
  }
  /** add_point_xyz_point_xyz_edge (graph_slam.cpp:168-180): g2o::EdgePointXYZ, measurement = p2 - p1 (never called upstream) */
  sslam::EdgeHandle add_point_xyz_point_xyz_edge(const sslam::VertexPointXYZ* v1_xyz, const sslam::VertexPointXYZ* v2_xyz,
                                                 const std::array<double, 3>& xyz, const double* information_matrix) {
    return {check(sslam_graph_add_edge_point_point(graph.get(), v1_xyz->id(), v2_xyz->id(), xyz.data(), information_matrix))};
  }
  /** add_se3_plane_edge (graph_slam.hpp:73-75, commented out upstream; include/g2o/edge_se3_plane.hpp) */
  sslam::EdgeHandle add_se3_plane_edge(const sslam::VertexSE3* v_se3, const sslam::VertexPlane* v_plane,
                                       const std::array<double, 4>& plane_coeffs, const double* information_matrix) {
    return {check(sslam_graph_add_edge_se3_plane(graph.get(), v_se3->id(), v_plane->id(), plane_coeffs.data(), information_matrix))};
  }

  /** The robust kernel graph_slam.cpp:155,161 means to install on the landmark edges (g2o::RobustKernelDCS; the reference passes an
   *  uninitialised pointer, so the default here is "none").  phi > 0 switches it on for every landmark edge, 0 off. */
  void setRobustKernelDCS(double phi = 1.0) { check(sslam_graph_set_option(graph.get(), "robust_kernel_dcs", phi)); }

  /** perform graph optimization (graph_slam.cpp:182-219): false iff the graph has fewer than 10 edges */
  bool optimize(int max_iterations = 1024) {
    sslam_opt_stats st;
    const int rc = sslam_graph_optimize(graph.get(), max_iterations, &st);
    if (rc == SSLAM_ERR_TOO_FEW_EDGES) return false;
    if (rc < 0) throw std::runtime_error(std::string("sslam_graph_optimize: ") + sslam_last_error());
    last_stats = st;
    if (verbose_) {
      std::cout << "done\niterations: " << st.iterations << "\nchi2: (before)" << st.chi2_before << " -> (after)" << st.chi2_after
                << "\ntime: " << st.seconds << "[sec]" << std::endl;
    }
    return true;
  }

  /** computeLandmarkMarginals (graph_slam.cpp:221-234), the reference's own signature: vert_pairs_vec holds (hessianIndex,
   *  hessianIndex) pairs as built at semantic_graph_slam.cpp:186-191; afterwards spinv.block(r, c) is that block of H^-1. */
  bool computeLandmarkMarginals(sslam::SparseBlockMatrix& spinv, std::vector<std::pair<int, int>> vert_pairs_vec) {
    std::vector<int> rc2;
    for (auto& pr : vert_pairs_vec) { rc2.push_back(pr.first); rc2.push_back(pr.second); }
    std::vector<double> out(vert_pairs_vec.size() * 36);
    const int rc = sslam_graph_marginals_by_hessian_index(graph.get(), rc2.data(), (int)vert_pairs_vec.size(), out.data());
    if (rc < 0) { if (verbose_) std::cout << "not computing marginals " << std::endl; return false; }
    std::map<int, int> dim_of;   // hessian index -> vertex dimension
    for (auto& v : vertices_) { const int hi = v->hessianIndex(); if (hi >= 0) dim_of[hi] = v->dimension(); }
    spinv.clear();
    size_t o = 0;
    for (auto& pr : vert_pairs_vec) {
      const int dr = dim_of[pr.first], dc = dim_of[pr.second];
      sslam::MarginalBlock* bl = spinv.block(pr.first, pr.second, true);
      bl->rows = dr; bl->cols = dc;
      bl->v.assign(out.begin() + o, out.begin() + o + (size_t)dr * dc);
      o += (size_t)dr * dc;
    }
    if (verbose_) std::cout << "computed marginals " << std::endl;
    return true;
  }
  /** convenience form: 3x3 blocks for a list of landmark vertex ids */
  bool computeLandmarkMarginals(std::vector<std::array<double, 9>>& spinv, const std::vector<int>& landmark_vertex_ids) {
    std::vector<double> out(landmark_vertex_ids.size() * 9);
    const int rc = sslam_graph_marginals(graph.get(), landmark_vertex_ids.data(), (int)landmark_vertex_ids.size(), out.data());
    if (rc < 0) { if (verbose_) std::cout << "not computing marginals " << std::endl; return false; }
    spinv.resize(landmark_vertex_ids.size());
    for (size_t k = 0; k < spinv.size(); ++k) for (int q = 0; q < 9; ++q) spinv[k][q] = out[k * 9 + q];
    if (verbose_) std::cout << "computed marginals " << std::endl;
    return true;
  }

  /** save the pose graph (graph_slam.cpp:236-239), g2o text format; like g2o's save() the failure is reported, not thrown */
  bool save(const std::string& filename) {
    const int rc = sslam_graph_save_g2o(graph.get(), filename.c_str());
    if (rc < 0) std::cerr << "error : failed to save the graph: " << sslam_last_error() << std::endl;
    return rc >= 0;
  }

#ifdef SSLAM_WITH_EIGEN
  static sslam::Isometry from_eigen(const Eigen::Isometry3d& T) {
    sslam::Isometry I;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) I.R[r * 3 + c] = T.linear()(r, c); I.t[r] = T.translation()(r); }
    return I;
  }
  sslam::VertexSE3* add_se3_node(const Eigen::Isometry3d& pose) { return add_se3_node(from_eigen(pose)); }
  sslam::VertexPointXYZ* add_point_xyz_node(const Eigen::Vector3d& xyz) { return add_point_xyz_node(std::array<double, 3>{xyz[0], xyz[1], xyz[2]}); }
  sslam::EdgeHandle add_se3_edge(const sslam::VertexSE3* v1, const sslam::VertexSE3* v2, const Eigen::Isometry3d& rel, const Eigen::MatrixXd& info) {
    double W[36]; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) W[r * 6 + c] = info(r, c);
    return add_se3_edge(v1, v2, from_eigen(rel), W);
  }
  sslam::EdgeHandle add_se3_point_xyz_edge(const sslam::VertexSE3* v, const sslam::VertexPointXYZ* p, const Eigen::Vector3d& xyz, const Eigen::MatrixXd& info) {
    double W[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) W[r * 3 + c] = info(r, c);
    return add_se3_point_xyz_edge(v, p, std::array<double, 3>{xyz[0], xyz[1], xyz[2]}, W);
  }
#endif

 public:
  bool verbose_;
  std::shared_ptr<sslam_graph> graph;  // the optimiser handle (g2o::SparseOptimizer in the reference, graph_slam.hpp:147)
  sslam_opt_stats last_stats{};

 private:
  template <typename T> T* own(T* v) { vertices_.emplace_back(v); return v; }
  std::vector<std::unique_ptr<sslam::VertexHandle>> vertices_;   // the graph owns its vertices, as g2o's optimizer does
  static int check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("sslam: ") + sslam_last_error());
    return rc;
  }
};

}  // namespace ps_graph_slam

#endif
