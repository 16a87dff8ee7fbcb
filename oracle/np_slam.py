"""TEST INFRASTRUCTURE — CPU restatement of the reference's orchestrator tick and data association.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  Parity unpinned: the reference holds no
tests or golden vectors for this path, and it cannot be built here (ROS / g2o / PCL); this file restates

    src/ps_graph_slam/semantic_graph_slam.cpp:58-102   run
                                              :104-150  empty_keyframe_queue
                                              :152-179  empty_landmark_queue
                                              :181-205  getAndSetLandmarkCov
                                              :234-287  VIOCallback
    include/ps_graph_slam/keyframe_updater.hpp:41-65    KeyframeUpdater::update
    include/ps_graph_slam/data_association.h:75-389     find_matches, associate_lanmarks, map_a_new_lan, inserst_a_mapped_lan, ...
    src/ps_graph_slam/information_matrix_calculator.cpp:28-35
    include/ps_graph_slam/ros_utils.hpp:90-106          matrix2vector
    include/tools.h:18-135                              transformNormalsToWorld, transformPoseFromCameraToRobot
    src/ps_graph_slam/semantic_graph_slam.cpp:207-232   semantic_data_ass: matrix2vector(robot_pose) -> segmentallPointCloudData
                                                        (the frontend oracle, oracle_seg.c) -> find_matches

with rigid transforms as 4x4 double matrices (Eigen::Isometry3d), the association in float32 (Eigen::MatrixXf / VectorXf), the
optimiser and the marginals through the C oracle (oracle_graph.c).  The decisions taken on the reference's undefined behaviour
(node-less landmarks inside a frame, distance_min, uninitialised information weights) are the ones listed at the top of
semantic_slam_amd/csrc/sslam_slam.hip.
"""
from __future__ import annotations

import math
import numpy as np

from . import oracle as O

F = np.float32
FLT_MAX = F(3.402823466e+38)


# ---- rigid transforms -----------------------------------------------------------------------------------------------------------
def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mat_to_quat(R):
    """Eigen::Quaternion(Matrix3): trace branch / largest diagonal branch."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def tq_to_iso(tq):
    q = np.asarray(tq[3:7], float)
    q = q / np.linalg.norm(q)
    T = np.eye(4)
    T[:3, :3] = quat_to_mat(q)
    T[:3, 3] = tq[:3]
    return T


def iso_to_tq(T):
    q = mat_to_quat(T[:3, :3])
    q = q / np.linalg.norm(q)
    return np.concatenate([T[:3, 3], q])


def iso_inv(T):
    R = T[:3, :3].T
    o = np.eye(4)
    o[:3, :3] = R
    o[:3, 3] = -R @ T[:3, 3]
    return o


def matrix2vector(T):
    """ros_utils.hpp:90-106: Quaternionf of the float rotation, normalised; tf::Matrix3x3(q).getEulerYPR in double."""
    q = mat_to_quat(T[:3, :3]).astype(F)
    q = (q / F(math.sqrt(float(np.sum(q * q, dtype=F))))).astype(F)
    x, y, z, w = [float(v) for v in q]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    m00, m10, m20, m21, m22 = 1.0 - (yy + zz), xy + wz, xz - wy, yz + wx, 1.0 - (xx + yy)
    if abs(m20) >= 1:
        yaw = 0.0
        delta = math.atan2(m21, m22)
        pitch = math.pi / 2 if m20 < 0 else -math.pi / 2
        roll = delta
    else:
        pitch = -math.asin(m20)
        roll = math.atan2(m21 / math.cos(pitch), m22 / math.cos(pitch))
        yaw = math.atan2(m10 / math.cos(pitch), m00 / math.cos(pitch))
    return np.array([T[0, 3], T[1, 3], T[2, 3], roll, pitch, yaw], F)


def _rot_x(a):
    M = np.zeros((4, 4), F)
    M[0, 0] = 1; M[1, 1] = F(math.cos(a)); M[1, 2] = F(-math.sin(a)); M[2, 1] = F(math.sin(a)); M[2, 2] = F(math.cos(a)); M[3, 3] = 1
    return M


def _rot_z(a):
    M = np.zeros((4, 4), F)
    M[0, 0] = F(math.cos(a)); M[0, 1] = F(-math.sin(a)); M[1, 0] = F(math.sin(a)); M[1, 1] = F(math.cos(a)); M[2, 2] = 1; M[3, 3] = 1
    return M


def _mm(A, B):
    """float32 4x4 product with the plain left-to-right accumulation of an un-vectorised Eigen product"""
    C = np.zeros((4, 4), F)
    for r in range(4):
        for c in range(4):
            s = F(0)
            for k in range(4):
                s = F(s + F(A[r, k] * B[k, c]))
            C[r, c] = s
    return C


def _mv(T, v):
    o = np.zeros(4, F)
    for r in range(4):
        s = F(0)
        for k in range(4):
            s = F(s + F(T[r, k] * v[k]))
        o[r] = s
    return o


def transform_normals_to_world(pose6, cam_angle, quirks=True):
    """tools.h:18-102 (quirk B2: element (0,2) uses sin(pitch) where sin(roll) is meant)"""
    roll, pitch, yaw = float(pose6[3]), float(pose6[4]), float(pose6[5])
    T = np.zeros((4, 4), F)
    c, s = math.cos, math.sin
    T[0, 0] = F(c(yaw) * c(pitch))
    T[0, 1] = F(c(yaw) * s(pitch) * s(roll) - s(yaw) * c(roll))
    T[0, 2] = F(c(yaw) * s(pitch) * c(roll) + s(yaw) * (s(pitch) if quirks else s(roll)))
    T[1, 0] = F(s(yaw) * c(pitch))
    T[1, 1] = F(s(yaw) * s(pitch) * s(roll) + c(yaw) * c(roll))
    T[1, 2] = F(s(yaw) * s(pitch) * c(roll) - c(yaw) * s(roll))
    T[2, 0] = F(-s(pitch)); T[2, 1] = F(c(pitch) * s(roll)); T[2, 2] = F(c(pitch) * c(roll)); T[3, 3] = 1
    return _mm(_mm(_mm(T, _rot_z(-1.5708)), _rot_x(-1.5708)), _rot_x(-float(F(cam_angle))))


def transform_cam_to_robot(cam_angle):
    return _mm(_mm(_rot_z(-1.5708), _rot_x(-1.5708)), _rot_x(-float(F(cam_angle))))


# ---- data association -------------------------------------------------------------------------------------------------------------
def _inverse_lu3(A):
    """inverse of a 3x3 float32 matrix by partial-pivot LU and three unit right-hand sides (Eigen's path for MatrixXf::inverse())"""
    A = A.astype(F).copy()
    piv = [0, 1, 2]
    for c in range(3):
        p = c
        best = abs(A[piv[c], c])
        for r in range(c + 1, 3):
            if abs(A[piv[r], c]) > best:
                best = abs(A[piv[r], c]); p = r
        piv[c], piv[p] = piv[p], piv[c]
        d = A[piv[c], c]
        for r in range(c + 1, 3):
            f = F(A[piv[r], c] / d)
            A[piv[r], c] = f
            for k in range(c + 1, 3):
                A[piv[r], k] = F(A[piv[r], k] - F(f * A[piv[c], k]))
    inv = np.zeros((3, 3), F)
    for col in range(3):
        y = np.zeros(3, F)
        for r in range(3):
            s = F(1.0 if piv[r] == col else 0.0)
            for k in range(r):
                s = F(s - F(A[piv[r], k] * y[k]))
            y[r] = s
        for r in (2, 1, 0):
            s = y[r]
            for k in range(r + 1, 3):
                s = F(s - F(A[piv[r], k] * inv[k, col]))
            inv[r, col] = F(s / A[piv[r], r])
    return inv


def mahalanobis(sigma, q, z):
    Q = sigma.astype(F).copy()
    for k in range(3):
        Q[k, k] = F(Q[k, k] + F(q))
    inv = _inverse_lu3(Q)
    rv = [F(F(F(z[0] * inv[0, c]) + F(z[1] * inv[1, c])) + F(z[2] * inv[2, c])) for c in range(3)]
    return F(F(F(rv[0] * z[0]) + F(rv[1] * z[1])) + F(rv[2] * z[2]))


class DataAssociation:
    """data_association.h; landmarks are dicts {id, vertex, class_id, plane_type, pose, local_pose, covariance, normal, is_new}"""

    def __init__(self, maha_dist_thres=0.5, eq_dist_thres=1.21, land_noise_low=0.5, use_maha_dist=True, use_eq_dist=False,
                 use_rtab_map_odom=False, keep_distance_min=False, quirks=True):
        self.maha, self.eq, self.q = maha_dist_thres, eq_dist_thres, F(land_noise_low)
        self.use_maha, self.use_eq, self.rtab = use_maha_dist, use_eq_dist, use_rtab_map_odom
        self.keep, self.quirks = keep_distance_min, quirks
        self.first_object = True
        self.landmarks = []

    def _views(self, obj, robot_pose, cam_angle):
        Tw = transform_normals_to_world(robot_pose, cam_angle, self.quirks)
        Tr = transform_cam_to_robot(cam_angle)
        pc = np.array([obj["pose"][0], obj["pose"][1], obj["pose"][2], 1.0], F)
        pw = _mv(Tw, pc)
        pw[0] = F(pw[0] + robot_pose[0])
        pw[1] = F(pw[1] + (F(float(robot_pose[1]) - 0.04) if self.rtab else robot_pose[1]))
        pw[2] = F(pw[2] + robot_pose[2])
        nw = _mv(Tw, np.asarray(obj["normal"], F))
        pr = _mv(Tr, pc)
        return pw, nw, pr

    def _record(self, obj, pw, nw, pr):
        cov = np.zeros((3, 3), F)
        cov[0, 0] = cov[1, 1] = cov[2, 2] = self.q
        return dict(class_id=obj["class_id"], plane_type=obj["plane_type"], pose=pw[:3].copy(), local_pose=pr[:3].copy(),
                    covariance=cov, normal=nw.copy(), vertex=-1, distance=-1.0)

    def _new(self, obj, pw, nw, pr):
        l = self._record(obj, pw, nw, pr)
        l["is_new"] = True
        l["id"] = len(self.landmarks)
        self.landmarks.append(dict(l, covariance=l["covariance"].copy()))
        return l

    def find_matches(self, objs, robot_pose, cam_angle, estimate_of):
        """estimate_of(landmark) -> float32 xyz the landmark's node currently holds"""
        out = []
        if self.first_object:
            for o in objs:
                out.append(self._new(o, *self._views(o, robot_pose, cam_angle)))
            if out:
                self.first_object = False
            return out
        dmin = FLT_MAX
        for o in objs:
            if not self.keep:
                dmin = FLT_MAX
            pw, nw, pr = self._views(o, robot_pose, cam_angle)
            found, best = False, -1
            for i, l in enumerate(self.landmarks):
                if l["class_id"] != o["class_id"] or l["plane_type"] != o["plane_type"]:
                    continue
                found = True
                h = estimate_of(l)
                z = np.array([F(pw[0] - h[0]), F(pw[1] - h[1]), F(pw[2] - h[2])], F)
                if self.use_maha:
                    dist = mahalanobis(l["covariance"], self.q, z)
                elif self.use_eq:
                    dist = F(np.sqrt(F(F(F(z[0] * z[0]) + F(z[1] * z[1])) + F(z[2] * z[2]))))
                else:
                    dist = F(0)
                if dist < dmin:
                    dmin, best = dist, i
            matched = False
            if found and best >= 0:
                if self.use_maha:
                    matched = not (float(dmin) > self.maha)
                elif self.use_eq:
                    matched = not (float(dmin) > self.eq)
            if matched:
                l = self._record(o, pw, nw, pr)
                l["is_new"] = False
                l["id"] = best
                l["vertex"] = self.landmarks[best]["vertex"]
            else:
                l = self._new(o, pw, nw, pr)
            l["distance"] = float(dmin) if (found and best >= 0) else -1.0
            out.append(l)
        return out


def _inverse3f(m):
    """Eigen's fixed-size 3x3 inverse: cofactors times 1/det (Matrix3f::inverse at semantic_graph_slam.cpp:170)"""
    m = m.astype(F).reshape(9)
    c00 = F(F(m[4] * m[8]) - F(m[5] * m[7])); c10 = F(F(m[5] * m[6]) - F(m[3] * m[8])); c20 = F(F(m[3] * m[7]) - F(m[4] * m[6]))
    det = F(F(F(m[0] * c00) + F(m[1] * c10)) + F(m[2] * c20))
    idet = F(F(1) / det)
    o = np.array([c00 * idet, F(F(m[2] * m[7]) - F(m[1] * m[8])) * idet, F(F(m[1] * m[5]) - F(m[2] * m[4])) * idet,
                  c10 * idet, F(F(m[0] * m[8]) - F(m[2] * m[6])) * idet, F(F(m[2] * m[3]) - F(m[0] * m[5])) * idet,
                  c20 * idet, F(F(m[1] * m[6]) - F(m[0] * m[7])) * idet, F(F(m[0] * m[4]) - F(m[1] * m[3])) * idet], F)
    return o.reshape(3, 3)


# ---- the tick -----------------------------------------------------------------------------------------------------------------------
class SemanticGraphSlam:
    def __init__(self, keyframe_delta_trans=0.5, keyframe_delta_angle=0.5, keyframe_delta_time=1.0, max_keyframes_per_update=10,
                 update_keyframes_using_detections=False, camera_angle_deg=0.0, const_stddev_x=0.0, const_stddev_q=0.0,
                 max_iterations=1024, seg_params=None, **da):
        self.dt, self.da_, self.dtime = keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time
        self.max_kf = max_keyframes_per_update
        self.using_det = update_keyframes_using_detections
        self.cam_angle = camera_angle_deg * (math.pi / 180)
        self.sx = const_stddev_x or 0.0667
        self.sq = const_stddev_q or 0.0667
        self.max_iterations = max_iterations
        self.assoc = DataAssociation(**da)
        self.object_detection_available = False
        self.first_key_added = False
        self.robot_pose = np.eye(4); self.prev_odom = np.eye(4); self.map2odom = np.eye(4); self.vio_pose = np.eye(4)
        self.queue, self.new_keyframes, self.keyframes = [], [], []
        self.latest_objects = None
        self.latest_cloud = None; self.latest_boxes = None
        self.seg_params = seg_params        # sslam_seg_params-layout struct for the frontend oracle (cloud path only)
        self.is_first, self.prev_keypose, self.prev_stamp, self.accum = True, np.eye(4), (0, 0), 0.0
        # the graph, as flat lists in the oracle's layout
        self.vtype, self.vfixed, self.est = [], [], []
        self.etype, self.evi, self.evj, self.meas, self.info = [], [], [], [], []
        self.last_stats = None

    # -- callbacks
    def set_segmented_objects(self, objs):
        self.object_detection_available = True
        self.latest_objects = [dict(o) for o in objs]

    def set_point_cloud(self, frame):
        """setPointCloudData (semantic_graph_slam.cpp:341-345); `frame`: synth.SynthFrame-like (cloud bytes + geometry)"""
        self.latest_cloud = frame

    def set_detected_objects(self, boxes):
        """setDetectedObjectInfo (semantic_graph_slam.cpp:353-357); `boxes`: the frame's structured box array"""
        self.object_detection_available = True
        self.latest_boxes = boxes
        self.latest_objects = None

    def _segment(self, kf):
        """semantic_data_ass (semantic_graph_slam.cpp:207-232): the keyframe's cloud and boxes through the frontend oracle, seen from
        matrix2vector(keyframe->robot_pose) (ros_utils.hpp:90-106)"""
        import dataclasses
        if self.seg_params is None:
            raise RuntimeError("a keyframe carries detection boxes but the oracle was built without seg_params")
        fr = dataclasses.replace(kf["cloud"], boxes=kf["boxes"], robot_pose=matrix2vector(kf["robot_pose"]).astype(F), cam_angle=float(F(self.cam_angle)))
        planes, _, _ = O.segment_frame(fr, self.seg_params)
        return [dict(pose=np.array(pl.centroid_cam[:], F), normal=np.array(pl.normal_d[:], F), class_id=int(pl.class_id), plane_type=int(pl.plane_type))
                for pl in planes]

    def _gate(self, odom, stamp):
        if self.is_first:
            self.is_first = False; self.prev_stamp = stamp; self.prev_keypose = odom
            return True
        delta = iso_inv(self.prev_keypose) @ odom
        dx = float(np.linalg.norm(delta[:3, 3]))
        da = math.acos(min(1.0, mat_to_quat(delta[:3, :3])[3] / np.linalg.norm(mat_to_quat(delta[:3, :3]))))
        dsec, dnsec = stamp[0] - self.prev_stamp[0], stamp[1] - self.prev_stamp[1]
        if dnsec < 0:
            dsec -= 1
        if dsec < self.dtime and dx < self.dt and da < self.da_:
            return False
        self.accum += dx; self.prev_keypose = odom; self.prev_stamp = stamp
        return True

    def vio(self, sec, nsec, odom_tq):
        odom = tq_to_iso(np.asarray(odom_tq, float))
        accept = self._gate(odom, (sec, nsec))
        reject = (not accept and not self.object_detection_available) if self.using_det else (not accept)
        if reject:
            if self.first_key_added:
                self.robot_pose = self.robot_pose @ (iso_inv(self.prev_odom) @ odom)
            self.vio_pose = odom; self.prev_odom = odom
            return False
        kf = dict(odom=odom, robot_pose=self.robot_pose.copy(), node=-1, objects=[], boxes=None, cloud=None)
        if self.object_detection_available:
            self.object_detection_available = False
            if self.latest_objects is not None:
                kf["objects"] = self.latest_objects
            else:                                    # getPointCloudData / getDetectedObjectInfo (:264-272)
                kf["boxes"], kf["cloud"] = self.latest_boxes, self.latest_cloud
        self.queue.append(kf)
        self.vio_pose = odom; self.prev_odom = odom
        return True

    # -- graph growth
    def _add_vertex(self, vtype, est):
        vid = len(self.vtype)
        self.vtype.append(vtype); self.vfixed.append(1 if (vtype == O.VT_SE3 and vid == 0) else 0)
        e = np.zeros(7); e[:len(est)] = est
        self.est.append(e)
        return vid

    def _add_edge(self, etype, i, j, meas, info):
        m = np.zeros(7); m[:len(meas)] = meas
        I = np.zeros(36); I[:info.size] = info.reshape(-1)
        self.etype.append(etype); self.evi.append(i); self.evj.append(j); self.meas.append(m); self.info.append(I)

    def _empty_keyframe_queue(self):
        if not self.queue:
            return False
        n = min(len(self.queue), self.max_kf)
        info = np.eye(6)
        info[:3, :3] /= self.sx
        info[3:, 3:] /= self.sq
        for i in range(n):
            kf = self.queue[i]
            self.new_keyframes.append(kf)
            kf["node"] = self._add_vertex(O.VT_SE3, iso_to_tq(kf["odom"]))
            if i == 0 and not self.keyframes:
                continue
            prev = self.keyframes[-1] if i == 0 else self.queue[i - 1]
            rel = iso_inv(prev["odom"]) @ kf["odom"]
            self._add_edge(O.ET_SE3, prev["node"], kf["node"], iso_to_tq(rel), info)
        del self.queue[:n]
        return True

    def _estimate_of(self, l):
        if l["vertex"] >= 0:
            return self.est[l["vertex"]][:3].astype(F)
        return l["pose"]

    def problem(self):
        return O.GraphProblem(self.vtype, self.vfixed, np.array(self.est), self.etype, self.evi, self.evj, np.array(self.meas), np.array(self.info))

    def run(self):
        if not self._empty_keyframe_queue():
            return False
        stats = dict(keyframes_added=len(self.new_keyframes), landmarks_added=0, landmarks_matched=0, landmark_edges_added=0,
                     optimized=False, marginals_ok=False, records=[])
        for kf in self.new_keyframes:
            if kf["boxes"] is not None and len(kf["boxes"]) > 0:
                kf["objects"] = self._segment(kf)
                kf["cloud"] = None
            if not kf["objects"]:
                continue
            rp = matrix2vector(kf["robot_pose"])
            cur = self.assoc.find_matches(kf["objects"], rp, F(self.cam_angle), self._estimate_of)
            stats["records"].append(cur)
            for l in cur:
                if l["is_new"]:
                    l["vertex"] = self._add_vertex(O.VT_POINT, l["pose"].astype(float))
                    l["is_new"] = False
                    self.assoc.landmarks[l["id"]]["vertex"] = l["vertex"]
                    stats["landmarks_added"] += 1
                else:
                    if l["vertex"] < 0:      # matched a landmark an earlier detection of the same frame created: it has its vertex by now
                        l["vertex"] = self.assoc.landmarks[l["id"]]["vertex"]
                    stats["landmarks_matched"] += 1
                inf = _inverse3f(l["covariance"]).astype(float)
                self._add_edge(O.ET_SE3_POINT, kf["node"], l["vertex"], l["local_pose"].astype(float), inf)
                stats["landmark_edges_added"] += 1
        self.keyframes += self.new_keyframes
        self.new_keyframes = []
        if len(self.etype) >= 10:                      # GraphSLAM::optimize (graph_slam.cpp:184-186)
            gp = self.problem()
            st = gp.optimize(self.max_iterations)
            self.est = [e.copy() for e in gp.est]
            stats["optimized"] = True
            stats["opt"] = st
            lms = self.assoc.landmarks
            if lms:
                try:
                    blocks = gp.marginals(np.array([l["vertex"] for l in lms], np.int32)).reshape(-1, 3, 3)
                    for l, b in zip(lms, blocks):
                        l["covariance"] = b.astype(F)
                    stats["marginals_ok"] = True
                except RuntimeError:
                    pass
            else:
                stats["marginals_ok"] = True
            last = self.keyframes[-1]
            self.robot_pose = tq_to_iso(self.est[last["node"]])
            self.map2odom = self.robot_pose @ iso_inv(last["odom"])
        self.first_key_added = True
        self.last_stats = stats
        return True
