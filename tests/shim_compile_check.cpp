// Compile-and-link check of the two C++ shims and, on a GPU box, an end-to-end run of both through the reference's own
// method names (GraphSLAM::optimize / computeLandmarkMarginals, point_cloud_segmentation::segmentallPointCloudData).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../include/ps_graph_slam_amd/graph_slam.hpp"
#include "../include/planar_segmentation_amd/point_cloud_segmentation.hpp"
#include "../include/ps_graph_slam_amd/semantic_graph_slam.hpp"

int main(int argc, char** argv) {
  ps_graph_slam::GraphSLAM slam(false);
  std::vector<sslam::VertexSE3*> nodes;                 // raw pointers owned by the graph, like g2o::VertexSE3*
  double W[36] = {0};
  for (int k = 0; k < 6; ++k) W[k * 7] = k < 3 ? 150.0 : 1e5;
  for (int i = 0; i < 12; ++i) {
    sslam::Isometry T = sslam::Isometry::Identity();
    T.t[0] = 0.5 * i + 0.01 * std::sin(3.0 * i);
    nodes.push_back(slam.add_se3_node(T));
    if (i > 0) { sslam::Isometry rel = sslam::Isometry::Identity(); rel.t[0] = 0.5; slam.add_se3_edge(nodes[i - 1], nodes[i], rel, W); }
  }
  // two landmarks seen from three poses each (semantic_graph_slam.cpp:160-174)
  double Wl[9] = {2.5, 0, 0, 0, 2.5, 0, 0, 0, 2.5};
  sslam::VertexPointXYZ* lm[2] = {slam.add_point_xyz_node({1.0, 1.0, 0.5}), slam.add_point_xyz_node({4.0, -1.0, 0.2})};
  for (int l = 0; l < 2; ++l)
    for (int i = 3 * l + 1; i < 3 * l + 4; ++i) {
      const std::array<double, 3> z = {(l ? 4.0 : 1.0) - 0.5 * i, l ? -1.0 : 1.0, l ? 0.2 : 0.5};
      slam.add_se3_point_xyz_edge(nodes[i], lm[l], z, Wl);
    }
  if (nodes[0]->hessianIndex() != -1 || nodes[1]->hessianIndex() != 0) { std::printf("hessian index wrong\n"); return 2; }
  if (sslam_device_count() < 1) { std::printf("shim ok (no GPU: compile/link/host-logic only)\n"); return 0; }
  if (!slam.optimize()) { std::printf("optimize returned false\n"); return 3; }
  const double x11 = nodes[11]->estimate().t[0];
  // getAndSetLandmarkCov, semantic_graph_slam.cpp:181-205
  sslam::SparseBlockMatrix spinv;
  std::vector<std::pair<int, int>> pairs;
  for (auto* l : lm) { l->unlockQuadraticForm(); pairs.push_back(std::make_pair(l->hessianIndex(), l->hessianIndex())); }
  if (!slam.computeLandmarkMarginals(spinv, pairs)) { std::printf("marginals failed\n"); return 5; }
  const sslam::MarginalBlock* c0 = spinv.block(lm[0]->hessianIndex(), lm[0]->hessianIndex());
  if (!c0 || c0->rows != 3 || !(c0->eval()(0, 0) > 0) || std::fabs(c0->eval()(0, 1) - c0->eval()(1, 0)) > 1e-9) { std::printf("marginal block wrong\n"); return 6; }
  if (slam.save("/nonexistent-dir/graph.g2o")) { std::printf("save() to a bad path reported success\n"); return 7; }
  std::printf("shim ok: chi2 %.3e -> %.3e, x[11] = %.6f, cov(l0)[0][0] = %.4e\n", slam.last_stats.chi2_before, slam.last_stats.chi2_after, x11, c0->eval()(0, 0));
  if (std::fabs(x11 - 5.5) > 1e-3) return 4;

  // frontend: the frame the caller wrote (tests/test_graph_gpu.py: a synthetic 640 x 480 cloud + boxes, little-endian:
  // int32 width, height, point_step, n_boxes; n_boxes x {int32 tl_x, tl_y, w, h}; float32 robot_pose[6], cam_angle; cloud bytes)
  if (argc < 3) { std::printf("frontend shim skipped (no frame file)\n"); return 0; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::printf("cannot open %s\n", argv[1]); return 8; }
  int32_t hdr[4];
  if (std::fread(hdr, 4, 4, f) != 4) return 8;
  const int w = hdr[0], h = hdr[1], step = hdr[2], nb = hdr[3];
  std::vector<ObjectInfo> info;
  for (int k = 0; k < nb; ++k) { int32_t b[4]; if (std::fread(b, 4, 4, f) != 4) return 8; info.push_back(ObjectInfo{"chair", 0.9f, b[0], b[1], b[2], b[3]}); }
  float pose7[7];
  if (std::fread(pose7, 4, 7, f) != 7) return 8;
  std::vector<uint8_t> cloud((size_t)w * h * step);
  if (std::fread(cloud.data(), 1, cloud.size(), f) != cloud.size()) return 8;
  std::fclose(f);
  point_cloud_segmentation seg(false);
  PointCloud2View view{cloud.data(), w, h, step, step * w, 0, 4, 8};
  std::vector<detected_object> objs = seg.segmentallPointCloudData({pose7[0], pose7[1], pose7[2], pose7[3], pose7[4], pose7[5]}, pose7[6], info, view);
  double cs = 0;   // checksum the caller recomputes from the Python mirror's planes
  for (auto& o : objs) cs += (double)o.normal_orientation[0] + 2.0 * o.normal_orientation[1] + 3.0 * o.normal_orientation[2] + 0.5 * o.normal_orientation[3] + o.num_points;
  std::printf("frontend shim ok: %zu planes checksum %.9e\n", objs.size(), cs);
  if ((int)objs.size() != std::atoi(argv[2])) return 9;
  {  // RANSAC per box + ICP of the frame against the boxes' own planes (the fixed point: T stays the identity, rms at the noise level)
    const std::vector<sslam_box_plane> rb = seg.ransacBoxes(0.01f, 50, 0.99, 5);
    std::vector<int32_t> box_plane(rb.size(), -1);
    std::vector<float> planes;
    int used = 0;
    for (size_t q = 0; q < rb.size(); ++q)
      if (rb[q].inliers > 500) { box_plane[q] = (int32_t)(planes.size() / 4); planes.insert(planes.end(), rb[q].coeff, rb[q].coeff + 4); ++used; }
    if (rb.empty() || used == 0) { std::printf("ransacBoxes found no plane\n"); return 15; }
    if ((int)seg.ransacBoxInliers(0).size() != rb[0].inliers) return 15;
    const std::vector<sslam_icp_result> ir = seg.icpBoxes(box_plane, planes.data(), (int)(planes.size() / 4), 3);
    if (ir.size() != 1 || ir[0].status != 0 || ir[0].used <= 0 || !(ir[0].rms < 0.01) || std::fabs(ir[0].T[0] - 1.0) > 5e-2 || std::fabs(ir[0].T[9]) > 0.2) {
      std::printf("icpBoxes: unexpected result (status %d used %d rms %g)\n", ir.empty() ? -99 : ir[0].status, ir.empty() ? 0 : ir[0].used, ir.empty() ? 0.0 : ir[0].rms);
      return 16;
    }
    std::printf("ransac / icp shim ok: %zu boxes, %d planes, icp rms %.2e over %d points\n", rb.size(), used, ir[0].rms, ir[0].used);
  }

  // orchestrator (semantic_graph_slam): the node's callbacks and loop on the same frame, seen from two keyframes 0.6 m apart
  semantic_graph_slam sgs;
  sgs.params().const_stddev_x = 0.00667; sgs.params().const_stddev_q = 0.00001;
  sgs.params().camera_angle_deg = pose7[6] * 180.0 / M_PI;   // ~camera_angle (semantic_graph_slam.cpp:24,29)
  sgs.init(false, seg.handle());
  std::vector<sslam_box> boxes;
  for (auto& o : info) boxes.push_back(sslam_box{o.tl_x, o.tl_y, o.width, o.height, SSLAM_CLASS_CHAIR, o.prob});
  for (int k = 0; k < 2; ++k) {
    sslam::Isometry odom = sslam::Isometry::Identity();
    odom.t[0] = 0.6 * k;
    sgs.setPointCloudData(cloud.data(), w, h, step, step * w, 0, 4, 8);
    sgs.setDetectedObjectInfo(boxes);
    if (!sgs.VIOCallback(k, 0, odom)) { std::printf("keyframe %d rejected\n", k); return 10; }
  }
  if (!sgs.run()) { std::printf("run() found no keyframes\n"); return 11; }
  std::vector<sslam_landmark> lms;
  sgs.getMappedLandmarks(lms);
  std::vector<std::pair<int, sslam::Isometry>> kfs;
  sgs.getKeyframes(kfs);
  const sslam_tick_stats& st = sgs.lastTick();
  std::printf("orchestrator shim ok: %d keyframes, %zu landmarks (%d new, %d matched), optimised %d, marginals %d\n", st.keyframes_added, lms.size(),
              st.landmarks_added, st.landmarks_matched, st.optimized, st.marginals_ok);
  // both keyframes carry the same cloud and boxes (seen from the orchestrator's own robot pose, so the plane count may differ from
  // the direct call above): every object of the first keyframe becomes a landmark, every object is an edge
  const bool enough = 1 + st.landmark_edges_added >= 10;   // GraphSLAM::optimize refuses fewer than 10 edges (graph_slam.cpp:184-186)
  if (kfs.size() != 2 || st.keyframes_added != 2 || st.landmark_edges_added != st.landmarks_added + st.landmarks_matched ||
      (int)lms.size() != st.landmarks_added || (enough && (!st.optimized || !st.marginals_ok))) return 12;
  if (seg.overflow_planes() || seg.overflow_candidate_boxes() || seg.overflow_region_boxes()) { std::printf("frontend tables overflowed\n"); return 13; }

  // legacy path (plane_segmentation::clusterAndSegmentAllPlanes / computeKmeans) through the shim: the scene the caller wrote
  // (int32 n; n x 3 float32 xyz; n x 3 float32 normals), identity transformation
  if (argc < 4) { std::printf("legacy shim skipped (no scene file)\n"); return 0; }
  f = std::fopen(argv[3], "rb");
  if (!f) { std::printf("cannot open %s\n", argv[3]); return 14; }
  int32_t n = 0;
  if (std::fread(&n, 4, 1, f) != 1 || n <= 0) return 14;
  std::vector<float> xyz((size_t)n * 3), nrm((size_t)n * 3);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(nrm.data(), 4, nrm.size(), f) != nrm.size()) return 14;
  std::fclose(f);
  const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const std::vector<std::array<float, 8>> rows = seg.clusterAndSegmentAllPlanes(xyz.data(), nrm.data(), n, T, 1);
  double rcs = 0;
  for (auto& r : rows) for (int k = 0; k < 8; ++k) rcs += (k + 1) * (double)r[k];
  std::printf("legacy shim ok: %zu rows rowsum %.9e\n", rows.size(), rcs);
  return 0;
}
