// Drop-in C++ shim: point_cloud_segmentation over the MI355X C-ABI (include/sslam.h).
//
// Mirrors reference include/planar_segmentation/point_cloud_segmentation.h:8-184: same class name and
// entry point `segmentallPointCloudData(robot_pose, cam_angle, object_info, point_cloud)`; ROS/PCL message
// types are replaced by plain structs with the same fields (ObjectInfo <- msg/ObjectInfo.msg:1-6,
// PointCloud2View <- the sensor_msgs::PointCloud2 members the reference reads at
// plane_segmentation.cpp:45-61, detected_object <- detected_object.h:14-24).
#ifndef PLANAR_SEGMENTATION_AMD_POINT_CLOUD_SEGMENTATION_HPP
#define PLANAR_SEGMENTATION_AMD_POINT_CLOUD_SEGMENTATION_HPP

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

  bool verbose_;

 private:
  sslam_seg* seg_ = nullptr;

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

  // the C-ABI handle (the orchestrator shim, ps_graph_slam_amd/semantic_graph_slam.hpp, borrows it for the tick's batched frontend pass)
  sslam_seg* handle() const { return seg_; }
};

#endif
