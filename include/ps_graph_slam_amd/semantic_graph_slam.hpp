// Drop-in C++ shim: the reference's orchestrator class, semantic_graph_slam (reference include/ps_graph_slam/semantic_graph_slam.h,
// src/ps_graph_slam/semantic_graph_slam.cpp), over the MI355X C-ABI (sslam_slam_* in include/sslam.h).
//
// Same method names and call pattern as the reference: the ROS node's callbacks call setPointCloudData / setDetectedObjectInfo /
// VIOCallback, its loop calls run(), the publishers read getRobotPose / getMap2OdomTrans / getMappedLandmarks / getKeyframes.
// ROS, Eigen, g2o and PCL types are replaced by plain ones: stamps are (sec, nsec), poses sslam::Isometry (graph_slam.hpp of this
// directory), the cloud is the PointCloud2 byte buffer with its layout, boxes are sslam_box, landmarks sslam_landmark.  The ROS
// private parameters read in semantic_graph_slam::init, KeyframeUpdater, InformationMatrixCalculator and data_association::init
// become the fields of sslam_slam_params (set them from ros::param in the node).
#ifndef PS_GRAPH_SLAM_AMD_SEMANTIC_GRAPH_SLAM_HPP
#define PS_GRAPH_SLAM_AMD_SEMANTIC_GRAPH_SLAM_HPP

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../sslam.h"
#include "graph_slam.hpp"

class semantic_graph_slam {
 public:
  semantic_graph_slam() { sslam_slam_default_params(&params_); }
  ~semantic_graph_slam() { if (h_) sslam_slam_destroy(h_); }
  semantic_graph_slam(const semantic_graph_slam&) = delete;
  semantic_graph_slam& operator=(const semantic_graph_slam&) = delete;

  // the ROS parameters (semantic_graph_slam.cpp:22-29, keyframe_updater.hpp:24-29, information_matrix_calculator.cpp:9-17,
  // data_association.h:49-55); change them before init()
  sslam_slam_params& params() { return params_; }

  // semantic_graph_slam::init (semantic_graph_slam.cpp:11-56).  seg: the frontend handle segmentallPointCloudData runs on
  // (point_cloud_segmentation of planar_segmentation_amd owns one); nullptr when objects arrive pre-segmented.
  void init(bool verbose, sslam_seg* seg = nullptr) {
    verbose_ = verbose;
    if (h_) sslam_slam_destroy(h_);
    h_ = sslam_slam_create(&params_, seg);
    if (!h_) throw std::runtime_error(std::string("sslam_slam_create: ") + sslam_last_error());
  }

  // semantic_graph_slam::run (semantic_graph_slam.cpp:58-102)
  bool run() {
    const int rc = sslam_slam_run(h_, &last_);
    if (rc < 0) throw std::runtime_error(std::string("sslam_slam_run: ") + sslam_last_error());
    return rc > 0;
  }
  const sslam_tick_stats& lastTick() const { return last_; }

  // VIOCallback (semantic_graph_slam.cpp:234-287); returns whether the sample became a keyframe
  bool VIOCallback(int32_t stamp_sec, int32_t stamp_nsec, const sslam::Isometry& odom) {
    double tq[7];
    sslam::isometry_to_tq(odom, tq);
    return check(sslam_slam_vio(h_, stamp_sec, stamp_nsec, tq), "sslam_slam_vio") > 0;
  }
  // setPointCloudData (semantic_graph_slam.cpp:341-345): sensor_msgs::PointCloud2::data + its layout
  void setPointCloudData(const uint8_t* data, int width, int height, int point_step, int row_step, int off_x, int off_y, int off_z) {
    check(sslam_slam_set_point_cloud(h_, data, width, height, point_step, row_step, off_x, off_y, off_z), "sslam_slam_set_point_cloud");
  }
  // setDetectedObjectInfo (semantic_graph_slam.cpp:353-357)
  void setDetectedObjectInfo(const std::vector<sslam_box>& object_info) {
    check(sslam_slam_set_detected_objects(h_, object_info.data(), (int)object_info.size()), "sslam_slam_set_detected_objects");
  }
  // extension: objects segmented elsewhere, for the next keyframe
  void setSegmentedObjects(const std::vector<sslam_plane>& objects) {
    check(sslam_slam_set_segmented_objects(h_, objects.data(), (int)objects.size()), "sslam_slam_set_segmented_objects");
  }

  // getRobotPose / getMap2OdomTrans (semantic_graph_slam.cpp:375-381)
  void getRobotPose(sslam::Isometry& robot_pose) const {
    double tq[7];
    check(sslam_slam_robot_pose(h_, tq), "sslam_slam_robot_pose");
    robot_pose = sslam::tq_to_isometry(tq);
  }
  void getMap2OdomTrans(sslam::Isometry& map2odom) const {
    double tq[7];
    check(sslam_slam_map2odom(h_, tq), "sslam_slam_map2odom");
    map2odom = sslam::tq_to_isometry(tq);
  }
  // getMappedLandmarks (semantic_graph_slam.cpp:366-368)
  void getMappedLandmarks(std::vector<sslam_landmark>& l_vec) const {
    const int n = check(sslam_slam_landmarks(h_, nullptr, 0), "sslam_slam_landmarks");
    l_vec.resize(n);
    if (n) check(sslam_slam_landmarks(h_, l_vec.data(), n), "sslam_slam_landmarks");
  }
  // getKeyframes (semantic_graph_slam.cpp:370-373): vertex id and optimised pose of every keyframe in the graph
  void getKeyframes(std::vector<std::pair<int, sslam::Isometry>>& keyframes) const {
    const int n = check(sslam_slam_keyframes(h_, nullptr, nullptr, 0), "sslam_slam_keyframes");
    std::vector<int32_t> ids(n);
    std::vector<double> est(7 * (size_t)n);
    if (n) check(sslam_slam_keyframes(h_, ids.data(), est.data(), n), "sslam_slam_keyframes");
    keyframes.clear();
    for (int i = 0; i < n; ++i) keyframes.emplace_back(ids[i], sslam::tq_to_isometry(&est[7 * (size_t)i]));
  }
  // saveGraph (semantic_graph_slam.cpp:391-394)
  void saveGraph(const std::string& save_graph_path) {
    check(sslam_graph_save_g2o(sslam_slam_graph(h_), save_graph_path.c_str()), "sslam_graph_save_g2o");
    std::cout << "saved the graph at " << save_graph_path << std::endl;
  }

 private:
  static int check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + sslam_last_error());
    return rc;
  }
  sslam_slam_params params_;
  sslam_slam* h_ = nullptr;
  sslam_tick_stats last_{};
  bool verbose_ = false;
};

#endif
