// Orchestrator tick without ROS: keyframe gate, keyframe / landmark queues, graph growth, optimisation,
// landmark marginals, map->odom, and data association on the device.
//
// Follows   src/ps_graph_slam/semantic_graph_slam.cpp:11-102 (init, run), :104-150 (empty_keyframe_queue),
//           :152-179 (empty_landmark_queue), :181-205 (getAndSetLandmarkCov), :207-232 (semantic_data_ass), :234-287 (VIOCallback),
//           :289-329 (addFirstPoseAndLandmark);
//           include/ps_graph_slam/keyframe_updater.hpp:41-65; include/ps_graph_slam/data_association.h:70-389;
//           src/ps_graph_slam/information_matrix_calculator.cpp:28-35; include/ps_graph_slam/ros_utils.hpp:90-106;
//           include/tools.h:104-135 (transformPoseFromCameraToRobot).
//
// What runs where.  The tick's heavy parts are the three C-ABI calls it makes into this same library -- the batched frontend pass
// over all keyframes of the tick (sslam_seg_segment_batch), sslam_graph_optimize and the landmark marginals -- all on the GPU.
// Data association is a single-workgroup kernel over the device-resident landmark table: the detections of a frame are resolved one
// after the other (a landmark created by detection j is a candidate for detection j+1, as in the reference where map_a_new_lan
// appends to landmarks_ inside the loop), the landmarks are scanned in parallel with a wave argmin.
//
// Deviations from the reference, all on undefined behaviour of the original:
//   * a landmark created earlier in the same frame has no graph node yet; the reference dereferences its uninitialised node
//     pointer (data_association.h:378 through :137); here its expected measurement is the position it was created at.
//   * distance_min is reset per detection unless reference_quirks bit 0 is set; neareast_landmarks_id
//     (uninitialised in the reference when no candidate beat distance_min) maps to "new landmark".
//   * InformationMatrixCalculator without ~use_const_inf_matrix reads members that are never initialised
//     (information_matrix_calculator.hpp:31-36); only the constant matrix is supported.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "../../include/sslam.h"
#include "sslam_common.hpp"
#include "sslam_math.hpp"

using namespace sslam;

namespace {

// ---- rigid transforms (Eigen::Isometry3d in the reference) as translation + unit quaternion -----------------------------------
Pose pose_identity() { return Pose{{0, 0, 0}, {0, 0, 0, 1}}; }
Quat qnormalized(Quat q) {
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return {q.x / n, q.y / n, q.z / n, q.w / n};
}
Pose compose(const Pose& a, const Pose& b) { return Pose{a.t + qrot(a.q, b.t), qnormalized(qmul(a.q, b.q))}; }
Pose inverse(const Pose& a) {
  const Quat qi = qconj(a.q);
  const Vec3 t = qrot(qi, a.t);
  return Pose{{-t.x, -t.y, -t.z}, qi};
}
Pose from_tq(const double* v) { return Pose{{v[0], v[1], v[2]}, qnormalized(Quat{v[3], v[4], v[5], v[6]})}; }
void to_tq(const Pose& p, double* v) {
  v[0] = p.t.x; v[1] = p.t.y; v[2] = p.t.z; v[3] = p.q.x; v[4] = p.q.y; v[5] = p.q.z; v[6] = p.q.w;
}

// ps_graph_slam::matrix2vector (ros_utils.hpp:90-106): float quaternion, normalised, then tf::Matrix3x3::getEulerYPR in double
void pose_to_vector6(const Pose& p, float out[6]) {
  float qx = (float)p.q.x, qy = (float)p.q.y, qz = (float)p.q.z, qw = (float)p.q.w;
  const float n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= n; qy /= n; qz /= n; qw /= n;
  const double x = qx, y = qy, z = qz, w = qw;
  // tf::Matrix3x3::setRotation
  const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  double yaw, pitch, roll;
  if (std::fabs(m20) >= 1) {   // gimbal lock branch of getEulerYPR
    yaw = 0;
    const double delta = std::atan2(m21, m22);
    if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; }
    else { pitch = -M_PI / 2.0; roll = delta; }
  } else {
    pitch = -std::asin(m20);
    roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
    yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
  }
  out[0] = (float)p.t.x; out[1] = (float)p.t.y; out[2] = (float)p.t.z;
  out[3] = (float)roll; out[4] = (float)pitch; out[5] = (float)yaw;
}

void mat4_mul(const float* A, const float* B, float* C) {
  float T[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      T[r * 4 + c] = s;
    }
  std::memcpy(C, T, sizeof T);
}

// semantic_tools::transformPoseFromCameraToRobot (tools.h:104-135)
void cam_to_robot(float cam_angle, float out[16]) {
  float rxc[16] = {0}, rxr[16] = {0}, rzr[16] = {0};
  const double a = -(double)cam_angle;
  rxc[0] = 1; rxc[5] = (float)std::cos(a); rxc[6] = (float)-std::sin(a); rxc[9] = (float)std::sin(a); rxc[10] = (float)std::cos(a); rxc[15] = 1;
  rxr[0] = 1; rxr[5] = (float)std::cos(-1.5708); rxr[6] = (float)-std::sin(-1.5708); rxr[9] = (float)std::sin(-1.5708); rxr[10] = (float)std::cos(-1.5708); rxr[15] = 1;
  rzr[0] = (float)std::cos(-1.5708); rzr[1] = (float)-std::sin(-1.5708); rzr[4] = (float)std::sin(-1.5708); rzr[5] = (float)std::cos(-1.5708); rzr[10] = 1; rzr[15] = 1;
  float M[16];
  mat4_mul(rzr, rxr, M);
  mat4_mul(M, rxc, out);
}

// ---- data association on the device ------------------------------------------------------------------------------------------
struct DetIn {           // what find_matches reads of a detected_object (detected_object.h:14-24)
  float pose[3];         // centroid in the camera frame
  float normal[4];
  int class_id, plane_type;
};
struct AssocArgs {
  float Tw[16];          // transformNormalsToWorld(robot_pose, cam_angle)
  float Tr[16];          // transformPoseFromCameraToRobot(cam_angle)
  float robot[3];        // robot position; y already carries the -0.04 of ~use_rtab_map_odom (data_association.h:335-338)
  float q;               // Q_ diagonal = land_noise_low
  double maha_thres, eq_thres;
  int use_maha, use_eq, first_object, keep_distance_min;
  int n_det;
};
struct LmTable {         // device-resident landmark list (structure of arrays)
  float* est;            // 3 per landmark: node->estimate() cast to float (landmarkMeasurementModel, data_association.h:375-380)
  float* cov;            // 9 per landmark
  int* kind;             // class_id | plane_type << 16
  int* count;            // number of landmarks
};

__device__ __forceinline__ void mat4_vec(const float* T, const float* v, float* o) {
  for (int r = 0; r < 4; ++r) {
    float s = 0;
    for (int k = 0; k < 4; ++k) s += T[r * 4 + k] * v[k];
    o[r] = s;
  }
}

// z^T (S + Q)^-1 z with the inverse formed the way Eigen forms it for a run-time sized matrix: partial-pivot LU, then the three
// columns of the identity solved one by one (data_association.h:158-168 works on Eigen::MatrixXf)
__device__ float mahalanobis3(const float* S, float q, const float* z) {
  float A[9];
  for (int k = 0; k < 9; ++k) A[k] = S[k];
  A[0] += q; A[4] += q; A[8] += q;
  int piv[3] = {0, 1, 2};
  for (int c = 0; c < 3; ++c) {
    int p = c;
    float best = fabsf(A[piv[c] * 3 + c]);
    for (int r = c + 1; r < 3; ++r) {
      const float v = fabsf(A[piv[r] * 3 + c]);
      if (v > best) { best = v; p = r; }
    }
    const int t = piv[c]; piv[c] = piv[p]; piv[p] = t;
    const float d = A[piv[c] * 3 + c];
    for (int r = c + 1; r < 3; ++r) {
      const float f = A[piv[r] * 3 + c] / d;
      A[piv[r] * 3 + c] = f;
      for (int k = c + 1; k < 3; ++k) A[piv[r] * 3 + k] -= f * A[piv[c] * 3 + k];
    }
  }
  float inv[9];
  for (int col = 0; col < 3; ++col) {
    float y[3];
    for (int r = 0; r < 3; ++r) {
      float s = piv[r] == col ? 1.0f : 0.0f;
      for (int k = 0; k < r; ++k) s -= A[piv[r] * 3 + k] * y[k];
      y[r] = s;
    }
    for (int r = 2; r >= 0; --r) {
      float s = y[r];
      for (int k = r + 1; k < 3; ++k) s -= A[piv[r] * 3 + k] * inv[k * 3 + col];
      inv[r * 3 + col] = s / A[piv[r] * 3 + r];
    }
  }
  float rv[3];
  for (int c = 0; c < 3; ++c) rv[c] = z[0] * inv[c] + z[1] * inv[3 + c] + z[2] * inv[6 + c];
  return rv[0] * z[0] + rv[1] * z[1] + rv[2] * z[2];
}

constexpr int kAssocThreads = 256;

// data_association::find_matches / associate_lanmarks / map_a_new_lan / inserst_a_mapped_lan (data_association.h:75-317).
// One workgroup; detections in order, landmarks in parallel.  The arg-min keeps the lowest landmark index among equal distances
// (the reference's strict `distance < distance_min` over ascending i).
__global__ __launch_bounds__(kAssocThreads) void k_associate(AssocArgs A, const DetIn* __restrict__ det, LmTable L, sslam_landmark* __restrict__ out) {
  __shared__ float s_d[kAssocThreads / 64];
  __shared__ int s_i[kAssocThreads / 64];
  __shared__ int s_any[kAssocThreads / 64];
  __shared__ int s_n;
  __shared__ float s_carry;   // distance_min carried across detections (quirk B5)
  const int tid = threadIdx.x;
  if (tid == 0) { s_n = *L.count; s_carry = 3.402823466e+38f; }
  __syncthreads();
  for (int j = 0; j < A.n_det; ++j) {
    const DetIn D = det[j];
    const float pc[4] = {D.pose[0], D.pose[1], D.pose[2], 1.0f};
    float pw[4], nw[4], pr[4];
    mat4_vec(A.Tw, pc, pw);                 // convertPoseToWorld (data_association.h:319-342)
    pw[0] += A.robot[0]; pw[1] += A.robot[1]; pw[2] += A.robot[2];
    mat4_vec(A.Tw, D.normal, nw);           // convertNormalsToWorld (:344-358)
    mat4_vec(A.Tr, pc, pr);                 // convertCamToRobot (:360-373)
    const int n = s_n;
    const int kind = D.class_id | (D.plane_type << 16);
    float dmin = A.keep_distance_min ? s_carry : 3.402823466e+38f;
    int imin = -1;
    bool any = false;
    if (!A.first_object) {
      for (int i = tid; i < n; i += kAssocThreads) {
        if (L.kind[i] != kind) continue;
        any = true;
        const float z[3] = {pw[0] - L.est[i * 3], pw[1] - L.est[i * 3 + 1], pw[2] - L.est[i * 3 + 2]};
        float dist = 0.0f;
        if (A.use_maha) dist = mahalanobis3(L.cov + i * 9, A.q, z);
        else if (A.use_eq) dist = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);   // semantic_tools::dist
        if (dist < dmin) { dmin = dist; imin = i; }
      }
    }
    // wave, then workgroup arg-min; ties -> lowest index
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(dmin, off);
      const int oi = __shfl_xor(imin, off);
      if (oi >= 0 && (od < dmin || (od == dmin && (imin < 0 || oi < imin)))) { dmin = od; imin = oi; }
    }
    const unsigned long long anyb = __ballot(any);
    if ((tid & 63) == 0) { s_d[tid >> 6] = dmin; s_i[tid >> 6] = imin; s_any[tid >> 6] = anyb != 0ull; }
    __syncthreads();
    if (tid == 0) {
      float bd = A.keep_distance_min ? s_carry : 3.402823466e+38f;
      int bi = -1;
      bool found = false;                   // found_nearest_neighbour: some landmark of the same type and plane type exists
      for (int w = 0; w < kAssocThreads / 64; ++w) {
        found = found || s_any[w];
        const float wd = s_d[w];
        const int wi = s_i[w];
        if (wi >= 0 && (wd < bd || (wd == bd && (bi < 0 || wi < bi)))) { bd = wd; bi = wi; }
      }
      if (A.keep_distance_min && bi >= 0) s_carry = bd;
      bool matched = false;
      if (found && bi >= 0) {
        if (A.use_maha) matched = !((double)bd > A.maha_thres);
        else if (A.use_eq) matched = !((double)bd > A.eq_thres);
      }
      sslam_landmark R;
      R.class_id = D.class_id; R.plane_type = D.plane_type;
      R.vertex = -1;
      for (int k = 0; k < 3; ++k) { R.pose[k] = pw[k]; R.local_pose[k] = pr[k]; }
      for (int k = 0; k < 4; ++k) R.normal[k] = nw[k];
      for (int k = 0; k < 9; ++k) R.covariance[k] = (k % 4 == 0) ? A.q : 0.0f;   // covariance = Q_ (:265, :304)
      R.distance = (found && bi >= 0) ? bd : -1.0f;
      if (matched) {
        R.is_new = 0; R.id = bi;
      } else {                              // map_a_new_lan: id = landmarks_.size(), appended at once (:259-269)
        R.is_new = 1; R.id = n;
        L.est[n * 3] = pw[0]; L.est[n * 3 + 1] = pw[1]; L.est[n * 3 + 2] = pw[2];
        for (int k = 0; k < 9; ++k) L.cov[n * 9 + k] = R.covariance[k];
        L.kind[n] = kind;
        s_n = n + 1;
        __threadfence_block();
      }
      out[j] = R;
    }
    __syncthreads();
  }
  if (tid == 0) *L.count = s_n;
}

// ---- host state ---------------------------------------------------------------------------------------------------------------
struct KeyFrame {
  int32_t sec = 0, nsec = 0;
  Pose odom = pose_identity(), robot_pose = pose_identity();
  double accum_distance = 0;
  int node = -1;
  std::vector<uint8_t> cloud;
  int width = 0, height = 0, point_step = 0, row_step = 0, off[3] = {0, 0, 0};
  std::vector<sslam_box> boxes;           // obj_info
  std::vector<sslam_plane> objects;       // pre-segmented (extension)
  bool presegmented = false;
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n, hipStream_t st, size_t keep) {   // grows geometrically, keeps the first `keep` elements
    if (n <= cap) return 0;
    size_t nc = cap ? cap : 256;
    while (nc < n) nc *= 2;
    T* q = nullptr;
    SSLAM_HIP_TRY(hipMalloc(&q, nc * sizeof(T)));
    if (p && keep) SSLAM_HIP_TRY(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
    SSLAM_HIP_TRY(hipStreamSynchronize(st));
    if (p) (void)hipFree(p);
    p = q; cap = nc;
    return 0;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct sslam_slam {
  sslam_slam_params P;
  sslam_seg* seg = nullptr;
  sslam_graph* graph = nullptr;
  double cam_angle = 0;
  // semantic_graph_slam state (semantic_graph_slam.h)
  bool object_detection_available = false, point_cloud_available = false, first_key_added = false;
  Pose robot_pose = pose_identity(), vio_pose = pose_identity(), prev_odom = pose_identity(), map2odom = pose_identity();
  std::deque<std::shared_ptr<KeyFrame>> keyframe_queue;
  std::vector<std::shared_ptr<KeyFrame>> new_keyframes, keyframes;
  KeyFrame latest;                        // point_cloud_msg_ / object_info_ / pre-segmented objects of the callbacks
  bool segmented_available = false;
  // KeyframeUpdater
  bool is_first = true;
  Pose prev_keypose = pose_identity();
  int32_t prev_sec = 0, prev_nsec = 0;
  double accum_distance = 0;
  // data_association
  bool first_object = true;
  std::vector<sslam_landmark> landmarks;
  // device side of the landmark list
  hipStream_t stream = nullptr;
  DevBuf<float> d_est, d_cov;
  DevBuf<int> d_kind, d_count;
  DevBuf<DetIn> d_det;
  DevBuf<sslam_landmark> d_out;
  sslam::PinnedScratch pin;               // page-locked staging of the table / detection / result copies
  bool table_dirty = true;                // host landmarks changed outside the kernel -> re-upload before the next association
  ~sslam_slam() {
    if (graph) sslam_graph_destroy(graph);
    if (stream) { (void)hipSetDevice(P.device); (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
  }
};

namespace {

int ensure_stream(sslam_slam* s) {
  if (s->stream) return 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible (%s); data association has no CPU fallback", hipGetErrorString(e));
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  return 0;
}

// host landmark list -> device table (after the marginals / estimates changed, or before the first association)
int upload_table(sslam_slam* s) {
  const size_t n = s->landmarks.size();
  int rc;
  if ((rc = s->d_est.reserve(3 * (n + 64), s->stream, 0))) return rc;
  if ((rc = s->d_cov.reserve(9 * (n + 64), s->stream, 0))) return rc;
  if ((rc = s->d_kind.reserve(n + 64, s->stream, 0))) return rc;
  if ((rc = s->d_count.reserve(1, s->stream, 0))) return rc;
  // staged page-locked: [est 3n | cov 9n | kind n | count]
  const size_t bytes = (13 * n + 1) * 4;
  std::vector<char> fallback;
  char* stage = s->pin.get(bytes);
  if (!stage) { fallback.resize(bytes); stage = fallback.data(); }
  float* est = reinterpret_cast<float*>(stage);
  float* cov = est + 3 * n;
  int* kind = reinterpret_cast<int*>(cov + 9 * n);
  int* cnt = kind + n;
  for (size_t i = 0; i < n; ++i) {
    const sslam_landmark& l = s->landmarks[i];
    if (l.vertex >= 0) {   // landmarkMeasurementModel: h = l.node->estimate().cast<float>()
      double p[3];
      if (sslam_graph_get_vertex(s->graph, l.vertex, p) < 0) return SSLAM_ERR_INVALID;
      for (int k = 0; k < 3; ++k) est[3 * i + k] = (float)p[k];
    } else {
      for (int k = 0; k < 3; ++k) est[3 * i + k] = l.pose[k];
    }
    for (int k = 0; k < 9; ++k) cov[9 * i + k] = l.covariance[k];
    kind[i] = l.class_id | (l.plane_type << 16);
  }
  *cnt = (int)n;
  if (n) {
    SSLAM_HIP_TRY(hipMemcpyAsync(s->d_est.p, est, 3 * n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(s->d_cov.p, cov, 9 * n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(s->d_kind.p, kind, n * sizeof(int), hipMemcpyHostToDevice, s->stream));
  }
  SSLAM_HIP_TRY(hipMemcpyAsync(s->d_count.p, cnt, sizeof(int), hipMemcpyHostToDevice, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  s->table_dirty = false;
  return 0;
}

// data_association::find_matches (data_association.h:75-95) -> landmark records of this frame's detections
int find_matches(sslam_slam* s, const sslam_plane* objs, int n, const float robot_pose[6], std::vector<sslam_landmark>& out) {
  out.clear();
  if (n <= 0) return 0;
  int rc = ensure_stream(s);
  if (rc) return rc;
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (s->table_dirty && (rc = upload_table(s))) return rc;
  const size_t nl = s->landmarks.size();
  if ((rc = s->d_est.reserve(3 * (nl + n), s->stream, 3 * nl))) return rc;
  if ((rc = s->d_cov.reserve(9 * (nl + n), s->stream, 9 * nl))) return rc;
  if ((rc = s->d_kind.reserve(nl + n, s->stream, nl))) return rc;
  if ((rc = s->d_det.reserve(n, s->stream, 0))) return rc;
  if ((rc = s->d_out.reserve(n, s->stream, 0))) return rc;
  // detections in, landmark records out: both through the page-locked staging buffer
  const size_t det_bytes = ((size_t)n * sizeof(DetIn) + 7) & ~(size_t)7, out_bytes = (size_t)n * sizeof(sslam_landmark);
  std::vector<char> fallback;
  char* stage = s->pin.get(det_bytes + out_bytes);
  if (!stage) { fallback.resize(det_bytes + out_bytes); stage = fallback.data(); }
  DetIn* det = reinterpret_cast<DetIn*>(stage);
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < 3; ++k) det[j].pose[k] = objs[j].centroid_cam[k];
    for (int k = 0; k < 4; ++k) det[j].normal[k] = objs[j].normal_d[k];
    det[j].class_id = objs[j].class_id; det[j].plane_type = objs[j].plane_type;
  }
  AssocArgs A;
  const float cam = (float)s->cam_angle;
  if ((rc = sslam_seg_transform(s->seg, robot_pose, cam, A.Tw)) < 0) return rc;
  cam_to_robot(cam, A.Tr);
  A.robot[0] = robot_pose[0];
  A.robot[1] = s->P.use_rtab_map_odom ? (float)((double)robot_pose[1] - 0.04) : robot_pose[1];
  A.robot[2] = robot_pose[2];
  A.q = (float)s->P.land_noise_low;
  A.maha_thres = s->P.maha_dist_thres; A.eq_thres = s->P.eq_dist_thres;
  A.use_maha = s->P.use_maha_dist; A.use_eq = s->P.use_eq_dist;
  A.first_object = s->first_object ? 1 : 0;
  A.keep_distance_min = s->P.reference_quirks & 1;
  A.n_det = n;
  LmTable L{s->d_est.p, s->d_cov.p, s->d_kind.p, s->d_count.p};
  SSLAM_HIP_TRY(hipMemcpyAsync(s->d_det.p, det, n * sizeof(DetIn), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_associate, dim3(1), dim3(kAssocThreads), 0, s->stream, A, s->d_det.p, L, s->d_out.p);
  SSLAM_HIP_TRY(hipGetLastError());
  SSLAM_HIP_TRY(hipMemcpyAsync(stage + det_bytes, s->d_out.p, out_bytes, hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  out.resize(n);
  memcpy(out.data(), stage + det_bytes, out_bytes);
  for (sslam_landmark& r : out) {
    if (r.is_new) {
      s->landmarks.push_back(r);                       // landmarks_.push_back(new_landmark) (:269)
    } else {
      r.vertex = s->landmarks[r.id].vertex;            // landmark.node = l.node (:309)
    }
  }
  if (s->first_object && !out.empty()) s->first_object = false;   // (:84-85)
  return 0;
}

// InformationMatrixCalculator::calc_information_matrix (information_matrix_calculator.cpp:28-35)
void odometry_information(const sslam_slam* s, double info[36]) {
  const double sx = s->P.const_stddev_x == 0 ? 0.0667 : s->P.const_stddev_x;
  const double sq = s->P.const_stddev_q == 0 ? 0.0667 : s->P.const_stddev_q;
  std::memset(info, 0, 36 * sizeof(double));
  for (int k = 0; k < 3; ++k) { info[k * 7] = 1.0 / sx; info[(k + 3) * 7] = 1.0 / sq; }
}

// empty_keyframe_queue (semantic_graph_slam.cpp:104-150)
int empty_keyframe_queue(sslam_slam* s) {
  if (s->keyframe_queue.empty()) return 0;
  const int n = std::min<int>((int)s->keyframe_queue.size(), s->P.max_keyframes_per_update);
  double info[36];
  odometry_information(s, info);
  int err = 0, used = 0;
  for (int i = 0; i < n && !err; ++i) {
    const std::shared_ptr<KeyFrame>& kf = s->keyframe_queue[i];
    double tq[7];
    to_tq(kf->odom, tq);
    kf->node = sslam_graph_add_vertex_se3(s->graph, tq, -1);
    used = i + 1;
    if (kf->node < 0) { err = kf->node; break; }      // the keyframe is dropped: it never reached the graph
    s->new_keyframes.push_back(kf);
    if (i == 0 && s->keyframes.empty()) continue;
    const std::shared_ptr<KeyFrame>& prev = i == 0 ? s->keyframes.back() : s->keyframe_queue[i - 1];
    const Pose rel = compose(inverse(prev->odom), kf->odom);
    double z[7];
    to_tq(rel, z);
    const int rc = sslam_graph_add_edge_se3(s->graph, prev->node, kf->node, z, info);
    if (rc < 0) err = rc;
  }
  // whatever was touched leaves the queue, also on an error: a second call must not add the same keyframes again
  s->keyframe_queue.erase(s->keyframe_queue.begin(), s->keyframe_queue.begin() + (err ? used : n));
  return err ? err : 1;
}

// Eigen's fixed-size 3x3 inverse (cofactors times 1/det), as `covariance.inverse()` of a Matrix3f evaluates (semantic_graph_slam.cpp:170)
void inverse3f(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c10 + m[2] * c20;
  const float id = 1.0f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c10 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c20 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// empty_landmark_queue (semantic_graph_slam.cpp:152-179)
int empty_landmark_queue(sslam_slam* s, std::vector<sslam_landmark>& q, const KeyFrame& kf, sslam_tick_stats* st) {
  for (sslam_landmark& l : q) {
    if (l.is_new) {
      const double p[3] = {l.pose[0], l.pose[1], l.pose[2]};
      l.vertex = sslam_graph_add_vertex_point(s->graph, p);
      if (l.vertex < 0) return l.vertex;
      l.is_new = 0;
      s->landmarks[l.id].vertex = l.vertex;            // assignLandmarkNode (:163-164)
      if (st) st->landmarks_added++;
    } else {
      // a landmark created by an earlier detection of the same frame had no vertex yet when find_matches copied it; the earlier
      // record of this queue has assigned it by now
      if (l.vertex < 0) l.vertex = s->landmarks[l.id].vertex;
      if (st) st->landmarks_matched++;
    }
    float inf[9];
    inverse3f(l.covariance, inf);
    double z[3], info[9];
    for (int k = 0; k < 3; ++k) z[k] = l.local_pose[k];
    for (int k = 0; k < 9; ++k) info[k] = inf[k];
    const int rc = sslam_graph_add_edge_se3_point(s->graph, kf.node, l.vertex, z, info);
    if (rc < 0) return rc;
    if (st) st->landmark_edges_added++;
  }
  return 0;
}

// getAndSetLandmarkCov (semantic_graph_slam.cpp:181-205)
int get_and_set_landmark_cov(sslam_slam* s, sslam_tick_stats* st) {
  const int n = (int)s->landmarks.size();
  if (n == 0) { if (st) st->marginals_ok = 1; return 0; }   // computeMarginals over an empty pair list
  // the reference hands (hessianIndex, hessianIndex) pairs to computeMarginals (:186-191); the vertex ids name the same blocks and
  // spare one scan of the vertex list per landmark
  std::vector<int> ids(n);
  for (int i = 0; i < n; ++i) ids[i] = s->landmarks[i].vertex;
  std::vector<double> blocks(9 * (size_t)n);
  const int rc = sslam_graph_marginals(s->graph, ids.data(), n, blocks.data());
  if (rc == SSLAM_ERR_NUMERIC || rc == SSLAM_ERR_INVALID) return 0;   // computeLandmarkMarginals returned false: covariances keep their values
  if (rc < 0) return rc;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 9; ++k) s->landmarks[i].covariance[k] = (float)blocks[9 * (size_t)i + k];   // setLandmarkCovs (:198-202)
  if (st) st->marginals_ok = 1;
  return 0;
}

// the segmented objects of every new keyframe that carries boxes: ONE batched frontend pass per tick (the reference runs
// segmentallPointCloudData keyframe by keyframe inside semantic_data_ass; the result does not depend on the association)
int segment_new_keyframes(sslam_slam* s) {
  std::vector<int> idx;
  for (size_t i = 0; i < s->new_keyframes.size(); ++i) {
    KeyFrame& kf = *s->new_keyframes[i];
    if (!kf.presegmented && !kf.boxes.empty()) idx.push_back((int)i);
  }
  if (idx.empty()) return 0;
  if (!s->seg) return set_error(SSLAM_ERR_INVALID, "a keyframe carries detection boxes but no frontend handle was given to sslam_slam_create");
  size_t a = 0;
  while (a < idx.size()) {   // runs of keyframes whose clouds share their geometry go through one call
    const KeyFrame& k0 = *s->new_keyframes[idx[a]];
    size_t b = a + 1;
    while (b < idx.size()) {
      const KeyFrame& k = *s->new_keyframes[idx[b]];
      if (k.width != k0.width || k.height != k0.height || k.point_step != k0.point_step || k.row_step != k0.row_step ||
          k.off[0] != k0.off[0] || k.off[1] != k0.off[1] || k.off[2] != k0.off[2]) break;
      ++b;
    }
    std::vector<sslam_frame> frames(b - a);
    int nbox = 0;
    for (size_t f = a; f < b; ++f) {
      KeyFrame& k = *s->new_keyframes[idx[f]];
      if (k.cloud.empty()) return set_error(SSLAM_ERR_INVALID, "keyframe with detection boxes has no point cloud");
      sslam_frame& F = frames[f - a];
      F.cloud = k.cloud.data(); F.boxes = k.boxes.data(); F.n_boxes = (int)k.boxes.size();
      pose_to_vector6(k.robot_pose, F.robot_pose);
      F.cam_angle = (float)s->cam_angle;
      nbox += F.n_boxes;
    }
    const int max_out = 64 * std::max(nbox, 1);
    std::vector<sslam_plane> planes(max_out);
    std::vector<int32_t> fr(max_out);
    const int np = sslam_seg_segment_batch(s->seg, frames.data(), (int)frames.size(), k0.width, k0.height, k0.point_step, k0.row_step,
                                           k0.off[0], k0.off[1], k0.off[2], planes.data(), max_out, fr.data());
    if (np < 0) return np;
    for (int k = 0; k < np; ++k) s->new_keyframes[idx[a + fr[k]]]->objects.push_back(planes[k]);
    a = b;
  }
  return 0;
}

}  // namespace

extern "C" {

void sslam_slam_default_params(sslam_slam_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->keyframe_delta_trans = 0.5; p->keyframe_delta_angle = 0.5; p->keyframe_delta_time = 1.0;
  p->max_keyframes_per_update = 10;
  p->first_lan[0] = 1.8; p->first_lan[1] = 0; p->first_lan[2] = 0.3;
  p->use_const_inf_matrix = 1;
  p->maha_dist_thres = 0.5; p->eq_dist_thres = 1.21; p->land_noise_low = 0.5; p->land_noise_high = 0.9;
  p->use_maha_dist = 1;
  p->max_iterations = 1024;
}

sslam_slam* sslam_slam_create(const sslam_slam_params* p, sslam_seg* seg) {
  sslam_slam_params P;
  if (p) P = *p; else sslam_slam_default_params(&P);
  if (!P.use_const_inf_matrix) {
    set_error(SSLAM_ERR_UNSUPPORTED, "use_const_inf_matrix = false reads uninitialised members in the reference (information_matrix_calculator.hpp:31-36)");
    return nullptr;
  }
  if (P.max_keyframes_per_update <= 0) { set_error(SSLAM_ERR_INVALID, "max_keyframes_per_update must be positive"); return nullptr; }
  sslam_slam* s = new sslam_slam();
  s->P = P;
  s->seg = seg;
  s->graph = sslam_graph_create(P.device);
  s->cam_angle = P.camera_angle_deg * (M_PI / 180);
  if (P.add_first_lan) {   // addFirstPoseAndLandmark (semantic_graph_slam.cpp:289-329)
    sslam_landmark l;
    std::memset(&l, 0, sizeof l);
    l.is_new = 1; l.id = 0; l.vertex = -1; l.class_id = SSLAM_CLASS_BUCKET; l.plane_type = 1;
    for (int k = 0; k < 3; ++k) l.pose[k] = l.local_pose[k] = (float)P.first_lan[k];
    l.normal[0] = -0.4f; l.normal[1] = 0.86f;
    l.covariance[0] = l.covariance[4] = l.covariance[8] = 0.1f;
    s->first_object = false;                    // addFirstLandmark (data_association.h:70-73)
    s->landmarks.push_back(l);
    auto kf = std::make_shared<KeyFrame>();
    kf->robot_pose = s->robot_pose;
    s->keyframe_queue.push_back(kf);
    if (empty_keyframe_queue(s) > 0) {
      std::vector<sslam_landmark> q{l};
      for (auto& k : s->new_keyframes) (void)empty_landmark_queue(s, q, *k, nullptr);
      for (auto& k : s->new_keyframes) s->keyframes.push_back(k);
      s->new_keyframes.clear();
    }
  }
  return s;
}

void sslam_slam_destroy(sslam_slam* s) { delete s; }

int sslam_slam_set_point_cloud(sslam_slam* s, const uint8_t* cloud, int width, int height, int point_step, int row_step, int off_x, int off_y, int off_z) {
  if (!s || !cloud || width <= 0 || height <= 0 || point_step <= 0 || row_step < width * point_step)
    return set_error(SSLAM_ERR_INVALID, "bad point cloud");
  s->point_cloud_available = true;
  KeyFrame& L = s->latest;
  L.cloud.assign(cloud, cloud + (size_t)row_step * height);
  L.width = width; L.height = height; L.point_step = point_step; L.row_step = row_step;
  L.off[0] = off_x; L.off[1] = off_y; L.off[2] = off_z;
  return 0;
}

int sslam_slam_set_detected_objects(sslam_slam* s, const sslam_box* boxes, int n) {
  if (!s || n < 0 || (n > 0 && !boxes)) return set_error(SSLAM_ERR_INVALID, "bad box list");
  s->object_detection_available = true;
  s->latest.boxes.assign(boxes, boxes + n);
  s->segmented_available = false;
  return 0;
}

int sslam_slam_set_segmented_objects(sslam_slam* s, const sslam_plane* objects, int n) {
  if (!s || n < 0 || (n > 0 && !objects)) return set_error(SSLAM_ERR_INVALID, "bad object list");
  s->object_detection_available = true;
  s->segmented_available = true;
  s->latest.objects.assign(objects, objects + n);
  return 0;
}

int sslam_slam_vio(sslam_slam* s, int32_t sec, int32_t nsec, const double odom_tq[7]) {
  if (!s || !odom_tq) return set_error(SSLAM_ERR_INVALID, "null argument");
  const Pose odom = from_tq(odom_tq);
  // KeyframeUpdater::update (keyframe_updater.hpp:41-65)
  bool accept;
  if (s->is_first) {
    s->is_first = false; s->prev_sec = sec; s->prev_nsec = nsec; s->prev_keypose = odom;
    accept = true;
  } else {
    const Pose delta = compose(inverse(s->prev_keypose), odom);
    const double dx = std::sqrt(dot(delta.t, delta.t));
    // Eigen::Quaterniond(delta.linear()).w(): the matrix -> quaternion conversion yields w >= 0, whatever the sign of the odometry
    // quaternions that went in (q and -q are the same rotation)
    const double da = std::acos(std::min(1.0, std::fabs(delta.q.w)));
    // ros::Duration::sec: whole seconds of the normalised difference (nsec part in [0, 1e9))
    int64_t dsec = (int64_t)sec - s->prev_sec, dnsec = (int64_t)nsec - s->prev_nsec;
    if (dnsec < 0) { dnsec += 1000000000; dsec -= 1; }
    if ((double)dsec < s->P.keyframe_delta_time && dx < s->P.keyframe_delta_trans && da < s->P.keyframe_delta_angle) {
      accept = false;
    } else {
      s->accum_distance += dx; s->prev_keypose = odom; s->prev_sec = sec; s->prev_nsec = nsec;
      accept = true;
    }
  }
  const bool reject = s->P.update_keyframes_using_detections ? (!accept && !s->object_detection_available) : !accept;
  if (reject) {   // semantic_graph_slam.cpp:239-262
    if (s->first_key_added) s->robot_pose = compose(s->robot_pose, compose(inverse(s->prev_odom), odom));
    s->vio_pose = odom; s->prev_odom = odom;
    return 0;
  }
  auto kf = std::make_shared<KeyFrame>();
  kf->sec = sec; kf->nsec = nsec; kf->odom = odom; kf->robot_pose = s->robot_pose; kf->accum_distance = s->accum_distance;
  // getPointCloudData / getDetectedObjectInfo (semantic_graph_slam.cpp:264-272): the latest cloud always, the boxes when a detection is pending
  s->point_cloud_available = false;
  if (s->object_detection_available) {
    s->object_detection_available = false;
    if (s->segmented_available) { kf->objects = s->latest.objects; kf->presegmented = true; s->segmented_available = false; }
    else {
      kf->boxes = s->latest.boxes;
      kf->cloud = s->latest.cloud;
      kf->width = s->latest.width; kf->height = s->latest.height; kf->point_step = s->latest.point_step; kf->row_step = s->latest.row_step;
      for (int k = 0; k < 3; ++k) kf->off[k] = s->latest.off[k];
    }
  }
  s->keyframe_queue.push_back(kf);
  s->vio_pose = odom; s->prev_odom = odom;
  return 1;
}

int sslam_slam_run(sslam_slam* s, sslam_tick_stats* st) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null handle");
  sslam_tick_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof *st);
  // on an error after the queue was touched the bookkeeping is completed before the error is returned (the keyframes that reached the
  // graph move to `keyframes`): the next call must neither add them again nor associate them a second time
  auto fail = [&](int code) {
    for (auto& kf : s->new_keyframes) { kf->cloud.clear(); kf->cloud.shrink_to_fit(); s->keyframes.push_back(kf); }
    s->new_keyframes.clear();
    s->table_dirty = true;
    return code;
  };
  int rc = empty_keyframe_queue(s);
  if (rc < 0) return fail(rc);
  if (rc == 0) return 0;
  st->keyframes_added = (int)s->new_keyframes.size();
  double t0 = now_s();
  if ((rc = segment_new_keyframes(s)) < 0) return fail(rc);
  st->seconds_frontend = now_s() - t0;
  t0 = now_s();
  for (auto& kfp : s->new_keyframes) {   // semantic_graph_slam.cpp:62-70
    KeyFrame& kf = *kfp;
    if (kf.presegmented ? kf.objects.empty() : kf.boxes.empty()) continue;
    float rp[6];
    pose_to_vector6(kf.robot_pose, rp);
    std::vector<sslam_landmark> cur;
    if ((rc = find_matches(s, kf.objects.data(), (int)kf.objects.size(), rp, cur)) < 0) return fail(rc);
    if ((rc = empty_landmark_queue(s, cur, kf, st)) < 0) return fail(rc);
    kf.cloud.clear(); kf.cloud.shrink_to_fit();
  }
  st->seconds_association = now_s() - t0;
  for (auto& kf : s->new_keyframes) s->keyframes.push_back(kf);
  s->new_keyframes.clear();
  t0 = now_s();
  rc = sslam_graph_optimize(s->graph, s->P.max_iterations, &st->opt);
  st->seconds_optimize = now_s() - t0;
  if (rc < 0 && rc != SSLAM_ERR_TOO_FEW_EDGES && rc != SSLAM_ERR_NUMERIC) return rc;
  if (rc != SSLAM_ERR_TOO_FEW_EDGES) {   // GraphSLAM::optimize returns true whatever LM did (graph_slam.cpp:218)
    st->optimized = 1;
    t0 = now_s();
    if ((rc = get_and_set_landmark_cov(s, st)) < 0) return rc;
    st->seconds_marginals = now_s() - t0;
    s->table_dirty = true;               // estimates and covariances moved
    const KeyFrame& last = *s->keyframes.back();
    double tq[7];
    if ((rc = sslam_graph_get_vertex(s->graph, last.node, tq)) < 0) return rc;
    s->robot_pose = from_tq(tq);
    s->map2odom = compose(s->robot_pose, inverse(last.odom));
  }
  s->first_key_added = true;
  return 1;
}

int sslam_slam_robot_pose(const sslam_slam* s, double tq[7]) {
  if (!s || !tq) return set_error(SSLAM_ERR_INVALID, "null argument");
  to_tq(s->robot_pose, tq);
  return 0;
}
int sslam_slam_map2odom(const sslam_slam* s, double tq[7]) {
  if (!s || !tq) return set_error(SSLAM_ERR_INVALID, "null argument");
  to_tq(s->map2odom, tq);
  return 0;
}
int sslam_slam_landmarks(const sslam_slam* s, sslam_landmark* out, int max) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null handle");
  const int n = (int)s->landmarks.size();
  for (int i = 0; i < n && i < max && out; ++i) {
    out[i] = s->landmarks[i];
  }
  return n;
}
int sslam_slam_keyframes(const sslam_slam* s, int32_t* ids, double* est, int max) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null handle");
  const int n = (int)s->keyframes.size();
  for (int i = 0; i < n && i < max; ++i) {
    if (ids) ids[i] = s->keyframes[i]->node;
    if (est && sslam_graph_get_vertex(s->graph, s->keyframes[i]->node, est + 7 * (size_t)i) < 0) return SSLAM_ERR_INVALID;
  }
  return n;
}
sslam_graph* sslam_slam_graph(sslam_slam* s) { return s ? s->graph : nullptr; }

int sslam_slam_find_matches(sslam_slam* s, const sslam_plane* objects, int n, const float robot_pose[6], sslam_landmark* out) {
  if (!s || n < 0 || (n > 0 && (!objects || !out)) || !robot_pose) return set_error(SSLAM_ERR_INVALID, "null argument");
  std::vector<sslam_landmark> cur;
  const int rc = find_matches(s, objects, n, robot_pose, cur);
  if (rc < 0) return rc;
  for (int j = 0; j < n; ++j) out[j] = cur[j];
  return n;
}

}  // extern "C"
