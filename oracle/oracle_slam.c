/* TEST INFRASTRUCTURE -- the orchestrator tick and data association of the reference as ONE plain-C driver.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  Parity unpinned (the reference holds no tests or
 * golden vectors for this path and cannot be built here: ROS / g2o / PCL).  It restates, for the pre-segmented path of the tick,
 *
 *     src/ps_graph_slam/semantic_graph_slam.cpp:58-102   run
 *                                               :104-150  empty_keyframe_queue
 *                                               :152-179  empty_landmark_queue
 *                                               :181-205  getAndSetLandmarkCov
 *                                               :234-287  VIOCallback
 *     include/ps_graph_slam/keyframe_updater.hpp:41-65    KeyframeUpdater::update
 *     include/ps_graph_slam/data_association.h:75-389     find_matches and what it calls
 *     src/ps_graph_slam/information_matrix_calculator.cpp:28-35
 *     include/ps_graph_slam/ros_utils.hpp:90-106          matrix2vector
 *     include/tools.h:18-135                              transformNormalsToWorld, transformPoseFromCameraToRobot
 *
 * and is the same arithmetic, statement for statement, as oracle/np_slam.py (which stays the readable restatement and also holds the cloud
 * path of the tick); tests/test_oracle_slam.py replays the same run through both and demands identical graphs, estimates and covariances.
 * Why it exists: np_slam.py spends half of a tick in NumPy / Python bookkeeping, which made the CPU side of bench.py's tick comparison
 * slower than a C++ node would be.  Here nothing but C runs inside the timed tick: association, graph growth, og_optimize, og_marginals.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- the graph oracle's interface (oracle_graph.c) ---- */
typedef struct { int nv, ne; int *vtype, *vfixed; double *est; int *etype, *evi, *evj; double *meas, *info; } og_problem;
typedef struct { int iterations, trials; double chi2_before, chi2_after, lambda, seconds, seconds_linearize, seconds_solve; int status; } og_stats;
int og_optimize(og_problem *P, int max_iters, og_stats *st);
int og_marginals(const og_problem *P, const int *ids, int nids, double *out);
enum { VT_SE3 = 0, VT_POINT = 1 };
enum { ET_SE3 = 0, ET_SE3_POINT = 1 };

typedef float F;
#define FLT_MAX_F 3.402823466e+38f

typedef struct { float pose[3]; float normal[4]; int class_id, plane_type; } oslam_object;
typedef struct {
  double keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time;
  int max_keyframes_per_update, update_keyframes_using_detections;
  double camera_angle_rad, const_stddev_x, const_stddev_q;
  int max_iterations;
  double maha_dist_thres, eq_dist_thres;
  float land_noise_low;
  int use_maha_dist, use_eq_dist, use_rtab_map_odom, keep_distance_min, quirks;
} oslam_params;
typedef struct {
  int keyframes_added, landmarks_added, landmarks_matched, landmark_edges_added, optimized, marginals_ok;
  int iterations, trials;
  double chi2_after, seconds_optimize, seconds_marginals, seconds_association, seconds_total;
} oslam_tick_stats;

typedef struct { int class_id, plane_type, vertex, id, is_new; F pose[3], local_pose[3], cov[9], normal[4]; double distance; } lmk_t;
typedef struct { double odom[16], robot_pose[16]; int node, nobj; oslam_object *obj; } kf_t;

typedef struct oslam {
  oslam_params p;
  int object_detection_available, first_key_added, first_object;
  double robot_pose[16], prev_odom[16];
  /* keyframe gate */
  int is_first; double prev_keypose[16]; int prev_sec, prev_nsec;
  /* latest detections */
  oslam_object *latest; int nlatest, caplatest;
  /* queues */
  kf_t *queue; int nqueue, capqueue;
  int nkeyframes; double last_odom[16]; int last_node;
  lmk_t *lmk; int nlmk, caplmk;
  /* graph in the oracle's layout */
  og_problem G; int capv, cape;
} oslam;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* ---- rigid transforms, 4x4 row-major doubles (Eigen::Isometry3d) ---- */
static void eye4(double *T) { memset(T, 0, 16 * sizeof(double)); T[0] = T[5] = T[10] = T[15] = 1; }
static void quat_to_mat(const double *q, double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static void mat_to_quat(const double *T, double q[4]) { /* Eigen::Quaternion(Matrix3): trace branch / largest diagonal branch; T: 4x4 */
#define Rm(r, c) T[4 * (r) + (c)]
  const double t = Rm(0, 0) + Rm(1, 1) + Rm(2, 2);
  if (t > 0) {
    double s = sqrt(t + 1.0);
    const double w = 0.5 * s;
    s = 0.5 / s;
    q[0] = (Rm(2, 1) - Rm(1, 2)) * s; q[1] = (Rm(0, 2) - Rm(2, 0)) * s; q[2] = (Rm(1, 0) - Rm(0, 1)) * s; q[3] = w;
    return;
  }
  int i = 0;
  if (Rm(1, 1) > Rm(0, 0)) i = 1;
  if (Rm(2, 2) > Rm(i, i)) i = 2;
  const int j = (i + 1) % 3, k = (i + 2) % 3;
  double s = sqrt(Rm(i, i) - Rm(j, j) - Rm(k, k) + 1.0);
  q[0] = q[1] = q[2] = q[3] = 0;
  q[i] = 0.5 * s;
  s = 0.5 / s;
  q[3] = (Rm(k, j) - Rm(j, k)) * s;
  q[j] = (Rm(j, i) + Rm(i, j)) * s;
  q[k] = (Rm(k, i) + Rm(i, k)) * s;
#undef Rm
}
static double norm4(const double *q) { return sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); }
static void tq_to_iso(const double *tq, double *T) {
  double q[4], R[9];
  const double n = norm4(tq + 3);
  for (int k = 0; k < 4; ++k) q[k] = tq[3 + k] / n;
  quat_to_mat(q, R);
  eye4(T);
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = tq[r]; }
}
static void iso_to_tq(const double *T, double *tq) {
  double q[4];
  mat_to_quat(T, q);
  const double n = norm4(q);
  tq[0] = T[3]; tq[1] = T[7]; tq[2] = T[11];
  for (int k = 0; k < 4; ++k) tq[3 + k] = q[k] / n;
}
static void iso_inv(const double *T, double *o) {
  eye4(o);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[4 * r + c] = T[4 * c + r];
  for (int r = 0; r < 3; ++r) {   /* -(R^T t), numpy's matrix-vector product: sum over k in order */
    double s = 0;
    for (int k = 0; k < 3; ++k) s += o[4 * r + k] * T[4 * k + 3];
    o[4 * r + 3] = -s;
  }
}
static void mm4(const double *A, const double *B, double *C) {
  double t[16];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += A[4 * r + k] * B[4 * k + c]; t[4 * r + c] = s; }
  memcpy(C, t, sizeof t);
}

/* ros_utils.hpp:90-106: Quaternionf of the float rotation, normalised; tf::Matrix3x3(q).getEulerYPR in double */
static void matrix2vector(const double *T, F out[6]) {
  double qd[4];
  mat_to_quat(T, qd);
  F q[4];
  for (int k = 0; k < 4; ++k) q[k] = (F)qd[k];
  F ss = (F)(q[0] * q[0]);
  for (int k = 1; k < 4; ++k) ss = (F)(ss + (F)(q[k] * q[k]));
  const F nrm = (F)sqrt((double)ss);
  for (int k = 0; k < 4; ++k) q[k] = (F)(q[k] / nrm);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = x * x + y * y + z * z + w * w;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  double roll, pitch, yaw;
  if (fabs(m20) >= 1) {
    yaw = 0.0;
    const double delta = atan2(m21, m22);
    pitch = m20 < 0 ? M_PI / 2 : -M_PI / 2;
    roll = delta;
  } else {
    pitch = -asin(m20);
    roll = atan2(m21 / cos(pitch), m22 / cos(pitch));
    yaw = atan2(m10 / cos(pitch), m00 / cos(pitch));
  }
  out[0] = (F)T[3]; out[1] = (F)T[7]; out[2] = (F)T[11]; out[3] = (F)roll; out[4] = (F)pitch; out[5] = (F)yaw;
}

/* ---- float32 4x4 algebra with the left-to-right accumulation of an un-vectorised Eigen product ---- */
static void f_rot_x(double a, F *M) { memset(M, 0, 16 * sizeof(F)); M[0] = 1; M[5] = (F)cos(a); M[6] = (F)(-sin(a)); M[9] = (F)sin(a); M[10] = (F)cos(a); M[15] = 1; }
static void f_rot_z(double a, F *M) { memset(M, 0, 16 * sizeof(F)); M[0] = (F)cos(a); M[1] = (F)(-sin(a)); M[4] = (F)sin(a); M[5] = (F)cos(a); M[10] = 1; M[15] = 1; }
static void f_mm(const F *A, const F *B, F *C) {
  F t[16];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { F s = 0; for (int k = 0; k < 4; ++k) s = (F)(s + (F)(A[4 * r + k] * B[4 * k + c])); t[4 * r + c] = s; }
  memcpy(C, t, sizeof t);
}
static void f_mv(const F *T, const F *v, F *o) {
  for (int r = 0; r < 4; ++r) { F s = 0; for (int k = 0; k < 4; ++k) s = (F)(s + (F)(T[4 * r + k] * v[k])); o[r] = s; }
}
/* tools.h:18-102 (quirk B2: element (0,2) uses sin(pitch) where sin(roll) is meant) */
static void transform_normals_to_world(const F pose6[6], F cam_angle, int quirks, F *out) {
  const double roll = pose6[3], pitch = pose6[4], yaw = pose6[5];
  F T[16], A[16], B[16];
  memset(T, 0, sizeof T);
  T[0] = (F)(cos(yaw) * cos(pitch));
  T[1] = (F)(cos(yaw) * sin(pitch) * sin(roll) - sin(yaw) * cos(roll));
  T[2] = (F)(cos(yaw) * sin(pitch) * cos(roll) + sin(yaw) * (quirks ? sin(pitch) : sin(roll)));
  T[4] = (F)(sin(yaw) * cos(pitch));
  T[5] = (F)(sin(yaw) * sin(pitch) * sin(roll) + cos(yaw) * cos(roll));
  T[6] = (F)(sin(yaw) * sin(pitch) * cos(roll) - cos(yaw) * sin(roll));
  T[8] = (F)(-sin(pitch)); T[9] = (F)(cos(pitch) * sin(roll)); T[10] = (F)(cos(pitch) * cos(roll)); T[15] = 1;
  f_rot_z(-1.5708, A); f_mm(T, A, B);
  f_rot_x(-1.5708, A); f_mm(B, A, T);
  f_rot_x(-(double)cam_angle, A); f_mm(T, A, out);
}
static void transform_cam_to_robot(F cam_angle, F *out) {
  F A[16], B[16], Cc[16];
  f_rot_z(-1.5708, A); f_rot_x(-1.5708, B); f_mm(A, B, Cc);
  f_rot_x(-(double)cam_angle, A); f_mm(Cc, A, out);
}

/* inverse of a 3x3 float matrix by partial-pivot LU and three unit right-hand sides (Eigen's path for MatrixXf::inverse()) */
static void inverse_lu3(const F *Ain, F *inv) {
  F A[9];
  memcpy(A, Ain, sizeof A);
  int piv[3] = {0, 1, 2};
  for (int c = 0; c < 3; ++c) {
    int p = c;
    F best = fabsf(A[3 * piv[c] + c]);
    for (int r = c + 1; r < 3; ++r) if (fabsf(A[3 * piv[r] + c]) > best) { best = fabsf(A[3 * piv[r] + c]); p = r; }
    const int t = piv[c]; piv[c] = piv[p]; piv[p] = t;
    const F d = A[3 * piv[c] + c];
    for (int r = c + 1; r < 3; ++r) {
      const F f = (F)(A[3 * piv[r] + c] / d);
      A[3 * piv[r] + c] = f;
      for (int k = c + 1; k < 3; ++k) A[3 * piv[r] + k] = (F)(A[3 * piv[r] + k] - (F)(f * A[3 * piv[c] + k]));
    }
  }
  for (int col = 0; col < 3; ++col) {
    F y[3];
    for (int r = 0; r < 3; ++r) {
      F s = piv[r] == col ? 1.0f : 0.0f;
      for (int k = 0; k < r; ++k) s = (F)(s - (F)(A[3 * piv[r] + k] * y[k]));
      y[r] = s;
    }
    for (int r = 2; r >= 0; --r) {
      F s = y[r];
      for (int k = r + 1; k < 3; ++k) s = (F)(s - (F)(A[3 * piv[r] + k] * inv[3 * k + col]));
      inv[3 * r + col] = (F)(s / A[3 * piv[r] + r]);
    }
  }
}
static F mahalanobis(const F *sigma, F q, const F *z) {
  F Q[9], inv[9], rv[3];
  memcpy(Q, sigma, sizeof Q);
  for (int k = 0; k < 3; ++k) Q[4 * k] = (F)(Q[4 * k] + q);
  inverse_lu3(Q, inv);
  for (int c = 0; c < 3; ++c) rv[c] = (F)((F)((F)(z[0] * inv[c]) + (F)(z[1] * inv[3 + c])) + (F)(z[2] * inv[6 + c]));
  return (F)((F)((F)(rv[0] * z[0]) + (F)(rv[1] * z[1])) + (F)(rv[2] * z[2]));
}
/* Eigen's fixed-size 3x3 inverse: cofactors times 1/det (Matrix3f::inverse at semantic_graph_slam.cpp:170) */
static void inverse3f(const F *m, F *o) {
  const F c00 = (F)((F)(m[4] * m[8]) - (F)(m[5] * m[7])), c10 = (F)((F)(m[5] * m[6]) - (F)(m[3] * m[8])), c20 = (F)((F)(m[3] * m[7]) - (F)(m[4] * m[6]));
  const F det = (F)((F)((F)(m[0] * c00) + (F)(m[1] * c10)) + (F)(m[2] * c20));
  const F idet = (F)(1.0f / det);
  o[0] = (F)(c00 * idet); o[1] = (F)((F)((F)(m[2] * m[7]) - (F)(m[1] * m[8])) * idet); o[2] = (F)((F)((F)(m[1] * m[5]) - (F)(m[2] * m[4])) * idet);
  o[3] = (F)(c10 * idet); o[4] = (F)((F)((F)(m[0] * m[8]) - (F)(m[2] * m[6])) * idet); o[5] = (F)((F)((F)(m[2] * m[3]) - (F)(m[0] * m[5])) * idet);
  o[6] = (F)(c20 * idet); o[7] = (F)((F)((F)(m[1] * m[6]) - (F)(m[0] * m[7])) * idet); o[8] = (F)((F)((F)(m[0] * m[4]) - (F)(m[1] * m[3])) * idet);
}

/* ---- growth of the flat arrays ---- */
static void grow_v(oslam *S) {
  if (S->G.nv < S->capv) return;
  S->capv = S->capv ? 2 * S->capv : 256;
  S->G.vtype = realloc(S->G.vtype, S->capv * sizeof(int)); S->G.vfixed = realloc(S->G.vfixed, S->capv * sizeof(int));
  S->G.est = realloc(S->G.est, (size_t)S->capv * 7 * sizeof(double));
}
static void grow_e(oslam *S) {
  if (S->G.ne < S->cape) return;
  S->cape = S->cape ? 2 * S->cape : 512;
  S->G.etype = realloc(S->G.etype, S->cape * sizeof(int)); S->G.evi = realloc(S->G.evi, S->cape * sizeof(int)); S->G.evj = realloc(S->G.evj, S->cape * sizeof(int));
  S->G.meas = realloc(S->G.meas, (size_t)S->cape * 7 * sizeof(double)); S->G.info = realloc(S->G.info, (size_t)S->cape * 36 * sizeof(double));
}
static int add_vertex(oslam *S, int vtype, const double *est, int n) {
  grow_v(S);
  const int vid = S->G.nv++;
  S->G.vtype[vid] = vtype; S->G.vfixed[vid] = (vtype == VT_SE3 && vid == 0) ? 1 : 0;
  double *e = S->G.est + 7 * (size_t)vid;
  memset(e, 0, 7 * sizeof(double));
  memcpy(e, est, n * sizeof(double));
  return vid;
}
static void add_edge(oslam *S, int etype, int i, int j, const double *meas, int nm, const double *info, int ni) {
  grow_e(S);
  const int k = S->G.ne++;
  S->G.etype[k] = etype; S->G.evi[k] = i; S->G.evj[k] = j;
  memset(S->G.meas + 7 * (size_t)k, 0, 7 * sizeof(double)); memcpy(S->G.meas + 7 * (size_t)k, meas, nm * sizeof(double));
  memset(S->G.info + 36 * (size_t)k, 0, 36 * sizeof(double)); memcpy(S->G.info + 36 * (size_t)k, info, ni * sizeof(double));
}

/* ---- public interface ---- */
oslam *oslam_create(const oslam_params *p) {
  oslam *S = calloc(1, sizeof(oslam));
  S->p = *p;
  if (S->p.const_stddev_x == 0) S->p.const_stddev_x = 0.0667;
  if (S->p.const_stddev_q == 0) S->p.const_stddev_q = 0.0667;
  S->first_object = 1; S->is_first = 1;
  eye4(S->robot_pose); eye4(S->prev_odom); eye4(S->prev_keypose); eye4(S->last_odom);
  S->last_node = -1;
  return S;
}
void oslam_destroy(oslam *S) {
  if (!S) return;
  for (int i = 0; i < S->nqueue; ++i) free(S->queue[i].obj);
  free(S->queue); free(S->latest); free(S->lmk);
  free(S->G.vtype); free(S->G.vfixed); free(S->G.est); free(S->G.etype); free(S->G.evi); free(S->G.evj); free(S->G.meas); free(S->G.info);
  free(S);
}
/* setSegmentedObjects: the detections the next accepted keyframe carries */
void oslam_set_objects(oslam *S, const oslam_object *objs, int n) {
  S->object_detection_available = 1;
  if (n > S->caplatest) { S->caplatest = n + 16; S->latest = realloc(S->latest, S->caplatest * sizeof(oslam_object)); }
  if (n > 0) memcpy(S->latest, objs, n * sizeof(oslam_object));
  S->nlatest = n;
}
/* keyframe_updater.hpp:41-65 */
static int gate(oslam *S, const double *odom, int sec, int nsec) {
  if (S->is_first) { S->is_first = 0; S->prev_sec = sec; S->prev_nsec = nsec; memcpy(S->prev_keypose, odom, 16 * sizeof(double)); return 1; }
  double inv[16], delta[16], q[4];
  iso_inv(S->prev_keypose, inv); mm4(inv, odom, delta);
  const double dx = sqrt(delta[3] * delta[3] + delta[7] * delta[7] + delta[11] * delta[11]);
  mat_to_quat(delta, q);
  const double da = acos(fmin(1.0, q[3] / norm4(q)));
  int dsec = sec - S->prev_sec;
  const int dnsec = nsec - S->prev_nsec;
  if (dnsec < 0) dsec -= 1;
  if (dsec < S->p.keyframe_delta_time && dx < S->p.keyframe_delta_trans && da < S->p.keyframe_delta_angle) return 0;
  memcpy(S->prev_keypose, odom, 16 * sizeof(double)); S->prev_sec = sec; S->prev_nsec = nsec;
  return 1;
}
/* VIOCallback (semantic_graph_slam.cpp:234-287): 1 when the sample became a keyframe */
int oslam_vio(oslam *S, int sec, int nsec, const double *odom_tq) {
  double odom[16];
  tq_to_iso(odom_tq, odom);
  const int accept = gate(S, odom, sec, nsec);
  const int reject = S->p.update_keyframes_using_detections ? (!accept && !S->object_detection_available) : !accept;
  if (reject) {
    if (S->first_key_added) { double inv[16], d[16]; iso_inv(S->prev_odom, inv); mm4(inv, odom, d); mm4(S->robot_pose, d, S->robot_pose); }
    memcpy(S->prev_odom, odom, sizeof odom);
    return 0;
  }
  if (S->nqueue == S->capqueue) { S->capqueue = S->capqueue ? 2 * S->capqueue : 32; S->queue = realloc(S->queue, S->capqueue * sizeof(kf_t)); }
  kf_t *kf = &S->queue[S->nqueue++];
  memcpy(kf->odom, odom, sizeof odom); memcpy(kf->robot_pose, S->robot_pose, sizeof odom);
  kf->node = -1; kf->nobj = 0; kf->obj = NULL;
  if (S->object_detection_available) {
    S->object_detection_available = 0;
    kf->nobj = S->nlatest;
    if (kf->nobj > 0) { kf->obj = malloc(kf->nobj * sizeof(oslam_object)); memcpy(kf->obj, S->latest, kf->nobj * sizeof(oslam_object)); }
  }
  memcpy(S->prev_odom, odom, sizeof odom);
  return 1;
}

/* ---- data association (data_association.h) ---- */
static void views(const oslam *S, const oslam_object *o, const F *rp, F cam_angle, const F *Tw, const F *Tr, F *pw, F *nw, F *pr) {
  const F pc[4] = {o->pose[0], o->pose[1], o->pose[2], 1.0f};
  f_mv(Tw, pc, pw);
  pw[0] = (F)(pw[0] + rp[0]);
  pw[1] = (F)(pw[1] + (S->p.use_rtab_map_odom ? (F)((double)rp[1] - 0.04) : rp[1]));
  pw[2] = (F)(pw[2] + rp[2]);
  f_mv(Tw, o->normal, nw);
  f_mv(Tr, pc, pr);
}
static void record(const oslam *S, const oslam_object *o, const F *pw, const F *nw, const F *pr, lmk_t *l) {
  memset(l, 0, sizeof *l);
  l->class_id = o->class_id; l->plane_type = o->plane_type;
  for (int k = 0; k < 3; ++k) { l->pose[k] = pw[k]; l->local_pose[k] = pr[k]; }
  for (int k = 0; k < 4; ++k) l->normal[k] = nw[k];
  l->cov[0] = l->cov[4] = l->cov[8] = S->p.land_noise_low;
  l->vertex = -1; l->distance = -1.0;
}
static void push_lmk(oslam *S, const lmk_t *l) {
  if (S->nlmk == S->caplmk) { S->caplmk = S->caplmk ? 2 * S->caplmk : 64; S->lmk = realloc(S->lmk, S->caplmk * sizeof(lmk_t)); }
  S->lmk[S->nlmk++] = *l;
}
static void estimate_of(const oslam *S, const lmk_t *l, F *h) {
  if (l->vertex >= 0) { const double *e = S->G.est + 7 * (size_t)l->vertex; h[0] = (F)e[0]; h[1] = (F)e[1]; h[2] = (F)e[2]; }
  else { h[0] = l->pose[0]; h[1] = l->pose[1]; h[2] = l->pose[2]; }
}
/* find_matches: out[n] (one record per detection) */
static void find_matches(oslam *S, const oslam_object *objs, int n, const F *rp, F cam_angle, lmk_t *out) {
  F Tw[16], Tr[16], pw[4], nw[4], pr[4];
  transform_normals_to_world(rp, cam_angle, S->p.quirks, Tw);
  transform_cam_to_robot(cam_angle, Tr);
  const F q = S->p.land_noise_low;
  if (S->first_object) {
    for (int k = 0; k < n; ++k) {
      views(S, &objs[k], rp, cam_angle, Tw, Tr, pw, nw, pr);
      record(S, &objs[k], pw, nw, pr, &out[k]);
      out[k].is_new = 1; out[k].id = S->nlmk;
      push_lmk(S, &out[k]);
    }
    if (n > 0) S->first_object = 0;
    return;
  }
  F dmin = FLT_MAX_F;
  for (int k = 0; k < n; ++k) {
    const oslam_object *o = &objs[k];
    if (!S->p.keep_distance_min) dmin = FLT_MAX_F;
    views(S, o, rp, cam_angle, Tw, Tr, pw, nw, pr);
    int found = 0, best = -1;
    for (int i = 0; i < S->nlmk; ++i) {
      const lmk_t *l = &S->lmk[i];
      if (l->class_id != o->class_id || l->plane_type != o->plane_type) continue;
      found = 1;
      F h[3];
      estimate_of(S, l, h);
      const F z[3] = {(F)(pw[0] - h[0]), (F)(pw[1] - h[1]), (F)(pw[2] - h[2])};
      F dist;
      if (S->p.use_maha_dist) dist = mahalanobis(l->cov, q, z);
      else if (S->p.use_eq_dist) dist = sqrtf((F)((F)((F)(z[0] * z[0]) + (F)(z[1] * z[1])) + (F)(z[2] * z[2])));
      else dist = 0;
      if (dist < dmin) { dmin = dist; best = i; }
    }
    int matched = 0;
    if (found && best >= 0) {
      if (S->p.use_maha_dist) matched = !((double)dmin > S->p.maha_dist_thres);
      else if (S->p.use_eq_dist) matched = !((double)dmin > S->p.eq_dist_thres);
    }
    record(S, o, pw, nw, pr, &out[k]);
    if (matched) { out[k].is_new = 0; out[k].id = best; out[k].vertex = S->lmk[best].vertex; }
    else { out[k].is_new = 1; out[k].id = S->nlmk; push_lmk(S, &out[k]); }
    out[k].distance = (found && best >= 0) ? (double)dmin : -1.0;
  }
}

/* run (semantic_graph_slam.cpp:58-102): 1 when a tick ran */
int oslam_run(oslam *S, oslam_tick_stats *st) {
  oslam_tick_stats z;
  memset(&z, 0, sizeof z);
  if (st) *st = z;
  if (S->nqueue == 0) return 0;
  const double t_begin = now_s();
  /* empty_keyframe_queue (:104-150) */
  const int n = S->nqueue < S->p.max_keyframes_per_update ? S->nqueue : S->p.max_keyframes_per_update;
  double info[36];
  memset(info, 0, sizeof info);
  for (int k = 0; k < 3; ++k) { info[7 * k] = 1.0 / S->p.const_stddev_x; info[7 * (k + 3)] = 1.0 / S->p.const_stddev_q; }
  for (int i = 0; i < n; ++i) {
    kf_t *kf = &S->queue[i];
    double tq[7];
    iso_to_tq(kf->odom, tq);
    kf->node = add_vertex(S, VT_SE3, tq, 7);
    if (i == 0 && S->nkeyframes == 0) continue;
    const double *podom = i == 0 ? S->last_odom : S->queue[i - 1].odom;
    const int pnode = i == 0 ? S->last_node : S->queue[i - 1].node;
    double inv[16], rel[16];
    iso_inv(podom, inv); mm4(inv, kf->odom, rel);
    iso_to_tq(rel, tq);
    add_edge(S, ET_SE3, pnode, kf->node, tq, 7, info, 36);
  }
  z.keyframes_added = n;
  /* empty_landmark_queue (:152-179) through find_matches */
  const double t_assoc = now_s();
  for (int i = 0; i < n; ++i) {
    kf_t *kf = &S->queue[i];
    if (kf->nobj <= 0) continue;
    F rp[6];
    matrix2vector(kf->robot_pose, rp);
    lmk_t *cur = malloc(kf->nobj * sizeof(lmk_t));
    find_matches(S, kf->obj, kf->nobj, rp, (F)S->p.camera_angle_rad, cur);
    for (int k = 0; k < kf->nobj; ++k) {
      lmk_t *l = &cur[k];
      if (l->is_new) {
        const double e[3] = {l->pose[0], l->pose[1], l->pose[2]};
        l->vertex = add_vertex(S, VT_POINT, e, 3);
        l->is_new = 0;
        S->lmk[l->id].vertex = l->vertex;
        z.landmarks_added++;
      } else {
        if (l->vertex < 0) l->vertex = S->lmk[l->id].vertex;   /* matched a landmark an earlier detection of the same frame created */
        z.landmarks_matched++;
      }
      F inff[9];
      inverse3f(l->cov, inff);
      double inf[9], m[3];
      for (int q = 0; q < 9; ++q) inf[q] = inff[q];
      for (int q = 0; q < 3; ++q) m[q] = l->local_pose[q];
      add_edge(S, ET_SE3_POINT, kf->node, l->vertex, m, 3, inf, 9);
      z.landmark_edges_added++;
    }
    free(cur);
  }
  z.seconds_association = now_s() - t_assoc;
  /* the new keyframes join the map */
  memcpy(S->last_odom, S->queue[n - 1].odom, 16 * sizeof(double));
  S->last_node = S->queue[n - 1].node;
  S->nkeyframes += n;
  for (int i = 0; i < n; ++i) free(S->queue[i].obj);
  memmove(S->queue, S->queue + n, (S->nqueue - n) * sizeof(kf_t));
  S->nqueue -= n;
  if (S->G.ne >= 10) {   /* GraphSLAM::optimize (graph_slam.cpp:184-186) */
    og_stats os;
    memset(&os, 0, sizeof os);
    const double t0 = now_s();
    og_optimize(&S->G, S->p.max_iterations, &os);
    z.seconds_optimize = now_s() - t0;
    z.optimized = 1; z.iterations = os.iterations; z.trials = os.trials; z.chi2_after = os.chi2_after;
    if (S->nlmk > 0) {   /* getAndSetLandmarkCov (:181-205) */
      const double t1 = now_s();
      int *ids = malloc(S->nlmk * sizeof(int));
      double *blocks = malloc((size_t)S->nlmk * 9 * sizeof(double));
      for (int i = 0; i < S->nlmk; ++i) ids[i] = S->lmk[i].vertex;
      if (og_marginals(&S->G, ids, S->nlmk, blocks) == 0) {
        for (int i = 0; i < S->nlmk; ++i) for (int q = 0; q < 9; ++q) S->lmk[i].cov[q] = (F)blocks[9 * (size_t)i + q];
        z.marginals_ok = 1;
      }
      free(ids); free(blocks);
      z.seconds_marginals = now_s() - t1;
    } else z.marginals_ok = 1;
    tq_to_iso(S->G.est + 7 * (size_t)S->last_node, S->robot_pose);
  }
  S->first_key_added = 1;
  z.seconds_total = now_s() - t_begin;
  if (st) *st = z;
  return 1;
}

/* ---- read-back ---- */
void oslam_counts(const oslam *S, int *nv, int *ne, int *nkeyframes, int *nlandmarks) { *nv = S->G.nv; *ne = S->G.ne; *nkeyframes = S->nkeyframes; *nlandmarks = S->nlmk; }
void oslam_graph(const oslam *S, int *vtype, double *est, int *etype, int *evi, int *evj, double *meas, double *info) {
  memcpy(vtype, S->G.vtype, S->G.nv * sizeof(int)); memcpy(est, S->G.est, (size_t)S->G.nv * 7 * sizeof(double));
  memcpy(etype, S->G.etype, S->G.ne * sizeof(int)); memcpy(evi, S->G.evi, S->G.ne * sizeof(int)); memcpy(evj, S->G.evj, S->G.ne * sizeof(int));
  memcpy(meas, S->G.meas, (size_t)S->G.ne * 7 * sizeof(double)); memcpy(info, S->G.info, (size_t)S->G.ne * 36 * sizeof(double));
}
void oslam_landmarks(const oslam *S, int *vertex, int *class_id, float *pose, float *cov) {
  for (int i = 0; i < S->nlmk; ++i) {
    vertex[i] = S->lmk[i].vertex; class_id[i] = S->lmk[i].class_id;
    memcpy(pose + 3 * i, S->lmk[i].pose, 3 * sizeof(float)); memcpy(cov + 9 * i, S->lmk[i].cov, 9 * sizeof(float));
  }
}
void oslam_robot_pose(const oslam *S, double *T16) { memcpy(T16, S->robot_pose, 16 * sizeof(double)); }
