// Compile-and-link check of the two C++ shims (and, on a GPU box, a tiny end-to-end run).
#include <cstdio>
#include <cmath>
#include "../include/ps_graph_slam_amd/graph_slam.hpp"
#include "../include/planar_segmentation_amd/point_cloud_segmentation.hpp"

int main() {
  ps_graph_slam::GraphSLAM slam(false);
  std::vector<std::shared_ptr<sslam::VertexSE3>> nodes;
  double W[36] = {0};
  for (int k = 0; k < 6; ++k) W[k * 7] = k < 3 ? 150.0 : 1e5;
  for (int i = 0; i < 12; ++i) {
    sslam::Isometry T = sslam::Isometry::Identity();
    T.t[0] = 0.5 * i + 0.01 * std::sin(3.0 * i);
    nodes.push_back(slam.add_se3_node(T));
    if (i > 0) { sslam::Isometry rel = sslam::Isometry::Identity(); rel.t[0] = 0.5; slam.add_se3_edge(nodes[i - 1].get(), nodes[i].get(), rel, W); }
  }
  if (nodes[0]->hessianIndex() != -1 || nodes[1]->hessianIndex() != 0) { std::printf("hessian index wrong\n"); return 2; }
  if (sslam_device_count() < 1) { std::printf("shim ok (no GPU: compile/link/host-logic only)\n"); return 0; }
  if (!slam.optimize()) { std::printf("optimize returned false\n"); return 3; }
  const double x11 = nodes[11]->estimate().t[0];
  std::printf("shim ok: chi2 %.3e -> %.3e, x[11] = %.6f\n", slam.last_stats.chi2_before, slam.last_stats.chi2_after, x11);
  return std::fabs(x11 - 5.5) < 1e-6 ? 0 : 4;
}
