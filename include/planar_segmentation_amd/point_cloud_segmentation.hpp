// Drop-in C++ shim: point_cloud_segmentation over the MI355X C-ABI (include/sslam.h).
//
// Mirrors reference include/planar_segmentation/point_cloud_segmentation.h:8-184: same class name and
// entry point `segmentallPointCloudData(robot_pose, cam_angle, object_info, point_cloud)`; ROS/PCL message
// types are replaced by plain structs with the same fields (ObjectInfo <- msg/ObjectInfo.msg:1-6,
// PointCloud2View <- the sensor_msgs::PointCloud2 members the reference reads at
// plane_segmentation.cpp:45-61, detected_object <- detected_object.h:14-24).
#ifndef PLANAR_SEGMENTATION_AMD_POINT_CLOUD_SEGMENTATION_HPP
#define PLANAR_SEGMENTATION_AMD_POINT_CLOUD_SEGMENTATION_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sslam.h"

struct ObjectInfo {       // semantic_SLAM::ObjectInfo
  std::string type;
  float prob;
  int tl_x, tl_y, width, height;
};

struct PointCloud2View {  // the parts of sensor_msgs::PointCloud2 the reference touches
  const uint8_t* data;
  int width, height, point_step, row_step;
  int offset_x, offset_y, offset_z;   // fields[0..2].offset
};

struct detected_object {  // detected_object.h:14-24
  int id = 0;
  float prob = 0;
  float num_points = 0;
  std::string type;
  std::string plane_type;
  std::array<float, 3> pose{};
  std::array<float, 3> world_pose{};
  std::array<float, 4> normal_orientation{};
};

class point_cloud_segmentation {
 public:
  explicit point_cloud_segmentation(bool verbose, const sslam_seg_params* params = nullptr) : verbose_(verbose) {
    sslam_seg_params p;
    if (params) p = *params; else sslam_seg_default_params(&p);
    seg_ = sslam_seg_create(&p);
    std::cout << "pc segmentation constructor " << std::endl;
  }
  ~point_cloud_segmentation() { sslam_seg_destroy(seg_); std::cout << "pc segmentation destructor " << std::endl; }
  point_cloud_segmentation(const point_cloud_segmentation&) = delete;
  point_cloud_segmentation& operator=(const point_cloud_segmentation&) = delete;

  /** point_cloud_segmentation.h:105-181; robot_pose = (x, y, z, roll, pitch, yaw) */
  std::vector<detected_object> segmentallPointCloudData(const std::array<float, 6>& robot_pose, float cam_angle,
                                                        const std::vector<ObjectInfo>& object_info, const PointCloud2View& point_cloud) {
    static const char* kNames[] = {"other", "chair", "tvmonitor", "book", "keyboard", "laptop", "bucket", "car"};
    std::vector<sslam_box> boxes(object_info.size());
    for (size_t i = 0; i < boxes.size(); ++i) {
      int cls = SSLAM_CLASS_OTHER;
      for (int k = 1; k < 8; ++k) if (object_info[i].type == kNames[k]) cls = k;   // whitelist of :126-130
      boxes[i] = sslam_box{object_info[i].tl_x, object_info[i].tl_y, object_info[i].width, object_info[i].height, cls, object_info[i].prob};
    }
    std::vector<sslam_plane> planes(64 * (boxes.size() + 1));
    const int n = sslam_seg_segment(seg_, point_cloud.data, point_cloud.width, point_cloud.height, point_cloud.point_step, point_cloud.row_step,
                                    point_cloud.offset_x, point_cloud.offset_y, point_cloud.offset_z, boxes.data(), (int)boxes.size(),
                                    robot_pose.data(), cam_angle, planes.data(), (int)planes.size());
    if (n < 0) throw std::runtime_error(std::string("sslam_seg_segment: ") + sslam_last_error());
    // truncation is reported by the C-ABI, and so by the drop-in: planes beyond the output capacity, boxes whose candidate / region
    // tables filled up (the reference has no such limits -- std::vector grows; a frame that hits them is not the reference's result)
    sslam_seg_last_overflow(seg_, &overflow_planes_, &overflow_candidate_boxes_, &overflow_region_boxes_);
    if (overflow_planes_ || overflow_candidate_boxes_ || overflow_region_boxes_)
      std::cerr << "point_cloud_segmentation: frontend tables overflowed (planes dropped " << overflow_planes_ << ", boxes with a full candidate table "
                << overflow_candidate_boxes_ << ", with a full region table " << overflow_region_boxes_ << ")" << std::endl;
    std::vector<detected_object> out(n);
    for (int k = 0; k < n; ++k) {
      const sslam_plane& p = planes[k];
      detected_object& o = out[k];
      o.type = kNames[p.class_id]; o.prob = p.prob; o.num_points = p.num_points;
      o.plane_type = p.plane_type == 0 ? "horizontal" : "vertical";
      for (int q = 0; q < 3; ++q) { o.pose[q] = p.centroid_cam[q]; o.world_pose[q] = p.world_pose[q]; }
      for (int q = 0; q < 4; ++q) o.normal_orientation[q] = p.normal_d[q];
    }
    return out;
  }

  /** plane_segmentation::compute2DConvexHull (plane_segmentation.cpp:631-665): RANSAC plane (threshold 0.01, refined
   *  coefficients) -> pcl::ProjectInliers -> 2-D pcl::ConvexHull.  xyz = n x 3 floats; returns the hull points. */
  std::vector<std::array<float, 3>> compute2DConvexHull(const float* xyz, int n, uint64_t seed = 0) {
    float coeff[4];
    std::vector<int32_t> inl(n > 0 ? n : 1);
    const int ni = sslam_seg_ransac_plane(seg_, xyz, n, 0.01f, 50, 0.99, seed, coeff, inl.data(), (int)inl.size());
    if (ni < 0) throw std::runtime_error(std::string("sslam_seg_ransac_plane: ") + sslam_last_error());
    std::vector<std::array<float, 3>> out;
    if (ni < 3) return out;
    std::vector<float> proj((size_t)ni * 3);
    std::vector<int32_t> hull(ni);
    const int h = sslam_seg_convex_hull_2d(seg_, xyz, n, inl.data(), ni, coeff, proj.data(), hull.data(), ni, nullptr);
    if (h < 0) throw std::runtime_error(std::string("sslam_seg_convex_hull_2d: ") + sslam_last_error());
    out.resize(h);
    for (int k = 0; k < h; ++k) out[k] = {proj[3 * (size_t)hull[k]], proj[3 * (size_t)hull[k] + 1], proj[3 * (size_t)hull[k] + 2]};
    return out;
  }

  /** plane_segmentation::computeKmeans (plane_segmentation.cpp:524-535): cv::kmeans(points, k, TermCriteria(EPS + MAX_ITER, 10000,
   *  1e-4), 10 attempts, KMEANS_RANDOM_CENTERS) on n x dim float rows -> labels, centres (k x dim), compactness */
  double computeKmeans(const float* points, int n, int dim, int num_centroids, std::vector<int32_t>& labels, std::vector<float>& centroids,
                       uint64_t seed = 0) {
    labels.assign(n > 0 ? n : 1, 0);
    centroids.assign((size_t)num_centroids * dim, 0.f);
    double compactness = 0;
    const int rc = sslam_seg_kmeans(seg_, points, n, dim, num_centroids, seed, labels.data(), centroids.data(), &compactness);
    if (rc < 0) throw std::runtime_error(std::string("sslam_seg_kmeans: ") + sslam_last_error());
    labels.resize(n);
    return compactness;
  }

  /** plane_segmentation::clusterAndSegmentAllPlanes (plane_segmentation.cpp:261-497; dead code upstream): k-means on the normals
   *  (4 centres, plane_segmentation.h:41) -> centres within +-0.3 of the horizontal-plane normal seen from the camera
   *  (filterCentroids, :499-518) -> per centre a k-means on the plane distances (2 centres, plane_segmentation.h:42) -> clusters
   *  of more than 500 points (:418) -> compute2DConvexHull.  xyz / normals: n x 3 floats (NaN normals are dropped, :479-497);
   *  transformation_mat: 4 x 4 row-major.  Returns the 1 x 8 rows the reference builds: [hull x, y, z, normal x, y, z, distance, 0]. */
  std::vector<std::array<float, 8>> clusterAndSegmentAllPlanes(const float* xyz, const float* normals, int n, const float transformation_mat[16],
                                                               uint64_t seed = 0) {
    std::vector<float> P, N;
    for (int i = 0; i < n; ++i) {
      const float* q = normals + 3 * (size_t)i;
      if (q[0] != q[0] || q[1] != q[1] || q[2] != q[2]) continue;
      P.insert(P.end(), xyz + 3 * (size_t)i, xyz + 3 * (size_t)i + 3);
      N.insert(N.end(), q, q + 3);
    }
    const int m = (int)(N.size() / 3);
    std::vector<std::array<float, 8>> rows;
    if (m <= 10) return rows;
    std::vector<int32_t> labels, dl;
    std::vector<float> centers, dc;
    computeKmeans(N.data(), m, 3, 4, labels, centers, seed);
    float n_cam[3];                                                        // T^T (0, 0, 1, 0)  (:343-345)
    for (int d = 0; d < 3; ++d) n_cam[d] = transformation_mat[2 * 4 + d];
    for (int cid = 0; cid < 4; ++cid) {
      const float* c = &centers[3 * (size_t)cid];
      bool near = true;
      for (int d = 0; d < 3; ++d) near = near && (n_cam[d] - 0.3f < c[d]) && (c[d] < n_cam[d] + 0.3f);
      if (!near) continue;
      std::vector<float> Pi;
      for (int i = 0; i < m; ++i) if (labels[i] == cid) Pi.insert(Pi.end(), &P[3 * (size_t)i], &P[3 * (size_t)i] + 3);
      const int mi = (int)(Pi.size() / 3);
      if (mi < 2) continue;
      std::vector<float> dist(mi);
      for (int i = 0; i < mi; ++i) {                                       // (:383-391), float accumulation in the reference's order
        const float a = Pi[3 * (size_t)i] * c[0] + Pi[3 * (size_t)i + 1] * c[1];
        const float b = a + Pi[3 * (size_t)i + 2] * c[2];
        dist[i] = -b;
      }
      computeKmeans(dist.data(), mi, 1, 2, dl, dc, seed + 1 + cid);
      for (int d = 0; d < 2; ++d) {
        std::vector<float> Q;
        for (int i = 0; i < mi; ++i) if (dl[i] == d) Q.insert(Q.end(), &Pi[3 * (size_t)i], &Pi[3 * (size_t)i] + 3);
        if ((int)(Q.size() / 3) <= 500) continue;
        for (const auto& h : compute2DConvexHull(Q.data(), (int)(Q.size() / 3), seed))
          rows.push_back({h[0], h[1], h[2], c[0], c[1], c[2], dc[d], 0.f});
      }
    }
    return rows;
  }

  /** truncation counters of the last segmentallPointCloudData call (sslam_seg_last_overflow): all zero = complete result */
  int overflow_planes() const { return overflow_planes_; }
  int overflow_candidate_boxes() const { return overflow_candidate_boxes_; }
  int overflow_region_boxes() const { return overflow_region_boxes_; }

  bool verbose_;

 private:
  sslam_seg* seg_ = nullptr;
  int overflow_planes_ = 0, overflow_candidate_boxes_ = 0, overflow_region_boxes_ = 0;

 public:
  // cloud filters of the legacy path (plane_segmentation.cpp:557-629), on unorganised xyz float clouds
  std::vector<int32_t> distance_filter(const float* xyz, int n, double dmin = 0.3, double dmax = 3.0) {
    std::vector<int32_t> keep(n > 0 ? n : 1);
    const int m = sslam_seg_distance_filter(seg_, xyz, n, dmin, dmax, keep.data(), n);
    if (m < 0) throw std::runtime_error(std::string("sslam_seg_distance_filter: ") + sslam_last_error());
    keep.resize(m);
    return keep;
  }
  std::vector<std::array<float, 3>> downsamplePointcloud(const float* xyz, int n, float leaf = 0.1f) {
    std::vector<std::array<float, 3>> out(n > 0 ? n : 1);
    const int m = sslam_seg_voxel_grid(seg_, xyz, n, leaf, &out[0][0], nullptr, (int)out.size());
    if (m < 0) throw std::runtime_error(std::string("sslam_seg_voxel_grid: ") + sslam_last_error());
    out.resize(m);
    return out;
  }
  std::vector<int32_t> removeOutliers(const float* xyz, int n, int mean_k = 50, double stddev_mul = 1.0) {
    std::vector<int32_t> keep(n > 0 ? n : 1);
    const int m = sslam_seg_statistical_outlier_removal(seg_, xyz, n, mean_k, stddev_mul, keep.data(), n, nullptr);
    if (m < 0) throw std::runtime_error(std::string("sslam_seg_statistical_outlier_removal: ") + sslam_last_error());
    keep.resize(m);
    return keep;
  }
  // point-to-plane ICP of labelled points against planes (north_star; no counterpart in the reference): T = R row-major | t
  struct IcpResult { std::array<double, 12> T; double rms; int points; };
  IcpResult icpPointToPlane(const float* xyz, const int32_t* labels, int n, const float* planes, int n_planes, int iterations = 10,
                            const double* T0 = nullptr) {
    IcpResult r{};
    r.points = sslam_seg_icp_point_to_plane(seg_, xyz, labels, n, planes, n_planes, iterations, T0, r.T.data(), &r.rms);
    if (r.points < 0) throw std::runtime_error(std::string("sslam_seg_icp_point_to_plane: ") + sslam_last_error());
    return r;
  }

  // RANSAC plane of every accepted box of the last segmentallPointCloudData call (the crops are still on the device) and point-to-plane
  // ICP of the frame over those inliers: BASELINE.json configs[3] "RANSAC+ICP plane extraction".  The reference's only RANSAC is
  // compute2DConvexHull's pcl::SACSegmentation (plane_segmentation.cpp:639-647: threshold 0.01, 50 iterations, probability 0.99).
  std::vector<sslam_box_plane> ransacBoxes(float threshold = 0.01f, int max_iterations = 50, double probability = 0.99, uint64_t seed = 0) {
    std::vector<sslam_box_plane> out(4096);
    const int n = sslam_seg_ransac_boxes(seg_, threshold, max_iterations, probability, seed, out.data(), (int)out.size(), nullptr);
    if (n < 0) throw std::runtime_error(std::string("sslam_seg_ransac_boxes: ") + sslam_last_error());
    out.resize(std::min<size_t>((size_t)n, out.size()));
    return out;
  }
  std::vector<int32_t> ransacBoxInliers(int slot) {
    std::vector<int32_t> idx(640 * 480);
    const int n = sslam_seg_ransac_box_inliers(seg_, slot, idx.data(), (int)idx.size());
    if (n < 0) throw std::runtime_error(std::string("sslam_seg_ransac_box_inliers: ") + sslam_last_error());
    idx.resize(std::min<size_t>((size_t)n, idx.size()));
    return idx;
  }
  // box_plane[slot]: index into planes (n_planes x 4) or -1; one result per frame of the resident batch
  std::vector<sslam_icp_result> icpBoxes(const std::vector<int32_t>& box_plane, const float* planes, int n_planes, int iterations = 10, const double* T0 = nullptr) {
    std::vector<sslam_icp_result> out(1024);
    const int n = sslam_seg_icp_boxes(seg_, box_plane.data(), (int)box_plane.size(), planes, n_planes, iterations, T0, out.data(), (int)out.size(), nullptr);
    if (n < 0) throw std::runtime_error(std::string("sslam_seg_icp_boxes: ") + sslam_last_error());
    out.resize(std::min<size_t>((size_t)n, out.size()));
    return out;
  }

  // the C-ABI handle (the orchestrator shim, ps_graph_slam_amd/semantic_graph_slam.hpp, borrows it for the tick's batched frontend pass)
  sslam_seg* handle() const { return seg_; }
};

#endif
