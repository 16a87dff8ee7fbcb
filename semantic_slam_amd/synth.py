"""Synthetic input generators for the semantic_slam hot path.

The reference ships no data and no tests (its only demo input is a rosbag that is
not in the tree), so the graphs / clouds below are shaped from the reference's
runtime parameters:

* keyframe spacing, odometry information ``diag(1/sx*I3, 1/sq*I3)``
  (reference ``src/ps_graph_slam/information_matrix_calculator.cpp:28-35``,
  ``config/bucket_detector.yaml:10-12,26-27``),
* landmark information ``(land_noise*I)^-1`` (``include/ps_graph_slam/data_association.h:64-66``,
  ``src/ps_graph_slam/semantic_graph_slam.cpp:170``),
* initial estimates exactly as the orchestrator produces them: poses = raw
  integrated odometry (``semantic_graph_slam.cpp:120-121``), landmark = first
  observing pose (initial estimate) composed with the measurement (``:160-161``).

Everything is deterministic in ``seed`` (``numpy.random.default_rng``).
Quaternions are stored ``(qx, qy, qz, qw)`` like g2o's ``VERTEX_SE3:QUAT`` rows.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# ----------------------------------------------------------------------------
# small quaternion helpers (vectorised, float64)
# ----------------------------------------------------------------------------

def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Hamilton product, (x, y, z, w) layout, broadcasting over leading dims."""
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ], axis=-1)


def quat_conj(q: np.ndarray) -> np.ndarray:
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_rotate(q: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Rotate vector(s) v by unit quaternion(s) q."""
    qv = q[..., :3]
    w = q[..., 3:4]
    t = 2.0 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


def quat_from_rotvec(r: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(r, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1.0), 0.5)
    return np.concatenate([r * k, np.cos(half)], axis=-1)


def quat_from_matrix(R: np.ndarray) -> np.ndarray:
    """Single 3x3 rotation matrix -> (x,y,z,w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr],
    ])


def pose_compose(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a ∘ b for poses stored as [t(3), q(4)]."""
    t = a[..., :3] + quat_rotate(a[..., 3:], b[..., :3])
    q = quat_mul(a[..., 3:], b[..., 3:])
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return np.concatenate([t, q], axis=-1)


def pose_inverse(a: np.ndarray) -> np.ndarray:
    qi = quat_conj(a[..., 3:])
    return np.concatenate([-quat_rotate(qi, a[..., :3]), qi], axis=-1)


# ----------------------------------------------------------------------------
# graph generator  G(Np, Nl, seed)
# ----------------------------------------------------------------------------

ODOM_STDDEV_X = 0.00667   # config/bucket_detector.yaml:26
ODOM_STDDEV_Q = 0.00001   # config/bucket_detector.yaml:27
LAND_NOISE = 0.4          # config/bucket_detector.yaml:22


@dataclasses.dataclass
class SynthGraph:
    """A synthetic pose/landmark graph in the reference's vocabulary."""
    poses_true: np.ndarray      # [Np,7]
    poses_init: np.ndarray      # [Np,7]  raw integrated odometry
    lms_true: np.ndarray        # [Nl,3] (points) or [Nl,4] (planes n,d)
    lms_init: np.ndarray
    odom_ij: np.ndarray         # [Eo,2] int32 (pose index i, pose index j)
    odom_z: np.ndarray          # [Eo,7]
    odom_info: np.ndarray       # [Eo,6,6]
    lm_ij: np.ndarray           # [El,2] int32 (pose index, landmark index)
    lm_z: np.ndarray            # [El,3] or [El,4]
    lm_info: np.ndarray         # [El,3,3]
    landmark_kind: str          # "point" | "plane"

    @property
    def n_poses(self) -> int:
        return self.poses_init.shape[0]

    @property
    def n_landmarks(self) -> int:
        return self.lms_init.shape[0]


def odom_information() -> np.ndarray:
    """information_matrix_calculator.cpp:28-35 (const branch; note I/sigma, quirk B3)."""
    inf = np.eye(6)
    inf[:3, :3] /= ODOM_STDDEV_X
    inf[3:, 3:] /= ODOM_STDDEV_Q
    return inf


def plane_transform_to_local(pose: np.ndarray, plane_w: np.ndarray) -> np.ndarray:
    """X^-1 ∘ pi  with  T∘(n,d) = (R n, d - t·(R n))  (SURVEY A.4)."""
    inv = pose_inverse(pose)
    n = quat_rotate(inv[..., 3:], plane_w[..., :3])
    d = plane_w[..., 3] - np.sum(inv[..., :3] * n, axis=-1)
    return np.concatenate([n, d[..., None]], axis=-1)


def plane_transform_to_world(pose: np.ndarray, plane_l: np.ndarray) -> np.ndarray:
    n = quat_rotate(pose[..., 3:], plane_l[..., :3])
    d = plane_l[..., 3] - np.sum(pose[..., :3] * n, axis=-1)
    return np.concatenate([n, d[..., None]], axis=-1)


def make_graph(n_poses: int, n_landmarks: int, seed: int = 0, *, landmark_kind: str = "point",
               k_obs: int = 3, loop_every: int = 50, noise_scale: float = 1.0) -> SynthGraph:
    rng = np.random.default_rng(seed)
    Np, Nl = int(n_poses), int(n_landmarks)
    # --- trajectory: 3 laps of a closed Lissajous loop --------------------------------
    s = np.linspace(0.0, 6.0 * np.pi, Np, endpoint=False)
    pos = np.stack([12.0 * np.cos(s), 8.0 * np.sin(s), 1.0 + 0.2 * np.sin(3.0 * s)], axis=1)
    dpos = np.stack([-12.0 * np.sin(s), 8.0 * np.cos(s), 0.6 * np.cos(3.0 * s)], axis=1)
    yaw = np.arctan2(dpos[:, 1], dpos[:, 0])
    roll = rng.normal(0.0, 0.02, Np)
    pitch = rng.normal(0.0, 0.02, Np)
    quat = np.stack([quat_from_matrix(rpy_to_matrix(roll[i], pitch[i], yaw[i])) for i in range(Np)])
    poses_true = np.concatenate([pos, quat], axis=1)

    # --- landmarks in a 3 m wide band around the track ---------------------------------
    sl = rng.uniform(0.0, 2.0 * np.pi, Nl)
    centre = np.stack([12.0 * np.cos(sl), 8.0 * np.sin(sl)], axis=1)
    tang = np.stack([-12.0 * np.sin(sl), 8.0 * np.cos(sl)], axis=1)
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    normal2 = np.stack([-tang[:, 1], tang[:, 0]], axis=1)
    off = rng.uniform(-1.5, 1.5, Nl)
    lm_xy = centre + off[:, None] * normal2
    lm_z = rng.uniform(0.0, 1.5, Nl)
    lm_pts = np.concatenate([lm_xy, lm_z[:, None]], axis=1)

    # --- observations: each pose sees its k nearest landmarks (all within ~3 m at S/L) -------
    # (chunked brute force; Np*Nl <= 5e6 for the L config)
    obs_p, obs_l = [], []
    kk = min(k_obs, Nl)
    for c0 in range(0, Np, 1024):
        d2 = ((pos[c0:c0 + 1024, None, :] - lm_pts[None, :, :]) ** 2).sum(-1)
        idx = np.argpartition(d2, kk - 1, axis=1)[:, :kk]
        dsel = np.take_along_axis(d2, idx, axis=1)
        order = np.argsort(dsel, axis=1, kind="stable")
        idx = np.take_along_axis(idx, order, axis=1)
        dsel = np.take_along_axis(dsel, order, axis=1)
        obs_p.append(np.repeat(np.arange(c0, c0 + idx.shape[0]), kk))
        obs_l.append(idx.reshape(-1))
    obs_p = np.concatenate(obs_p).astype(np.int32)
    obs_l = np.concatenate(obs_l).astype(np.int32)
    # every landmark must be seen at least once (g2o drops edge-less vertices from the
    # active set; the generator avoids that corner): hand an unseen landmark to its nearest
    # pose in place of that pose's farthest observation, if that does not orphan another.
    cnt = np.bincount(obs_l, minlength=Nl)
    for l in np.nonzero(cnt == 0)[0]:
        order_p = np.argsort(((pos - lm_pts[l]) ** 2).sum(-1), kind="stable")
        for p in order_p:
            slot = p * kk + (kk - 1)
            if cnt[obs_l[slot]] > 1:
                cnt[obs_l[slot]] -= 1
                obs_l[slot] = l
                cnt[l] += 1
                break
    lm_ij = np.stack([obs_p, obs_l], axis=1)
    El = lm_ij.shape[0]

    # --- odometry edges (sequential) + explicit loop closures --------------------------
    i_seq = np.arange(Np - 1)
    pairs = [np.stack([i_seq, i_seq + 1], axis=1)]
    lap = Np // 3
    n_loops = Np // loop_every if loop_every > 0 else 0
    if n_loops > 0 and lap > 0:
        starts = rng.integers(0, Np - lap, n_loops)
        hops = rng.integers(1, 3, n_loops)
        tgt = np.minimum(starts + hops * lap, Np - 1)
        pairs.append(np.stack([starts, tgt], axis=1))
    odom_ij = np.concatenate(pairs, axis=0).astype(np.int32)
    Eo = odom_ij.shape[0]
    rel_true = pose_compose(pose_inverse(poses_true[odom_ij[:, 0]]), poses_true[odom_ij[:, 1]])
    n_t = rng.normal(0.0, 0.02 * noise_scale, (Eo, 3))
    n_r = rng.normal(0.0, 0.01 * noise_scale, (Eo, 3))
    noise = np.concatenate([n_t, quat_from_rotvec(n_r)], axis=1)
    odom_z = pose_compose(rel_true, noise)
    flip = odom_z[:, 6] < 0
    odom_z[flip, 3:] *= -1.0
    odom_info = np.broadcast_to(odom_information(), (Eo, 6, 6)).copy()

    # --- initial pose estimates: integrate the noisy sequential odometry ---------------
    poses_init = np.empty_like(poses_true)
    poses_init[0] = poses_true[0]
    for i in range(Np - 1):
        poses_init[i + 1] = pose_compose(poses_init[i], odom_z[i])

    # --- landmark measurements + initial estimates -------------------------------------
    lm_info = np.broadcast_to(np.eye(3) / LAND_NOISE, (El, 3, 3)).copy()
    pi = lm_ij[:, 0]
    li = lm_ij[:, 1]
    if landmark_kind == "point":
        lms_true = lm_pts
        z_true = quat_rotate(quat_conj(poses_true[pi, 3:]), lm_pts[li] - poses_true[pi, :3])
        lm_zm = z_true + rng.normal(0.0, 0.05 * noise_scale, (El, 3))
        lms_init = np.zeros((Nl, 3))
        seen = np.zeros(Nl, bool)
        for e in range(El):
            l = li[e]
            if not seen[l]:
                seen[l] = True
                lms_init[l] = poses_init[pi[e], :3] + quat_rotate(poses_init[pi[e], 3:], lm_zm[e])
        lms_init[~seen] = lm_pts[~seen]
    elif landmark_kind == "plane":
        nrm = rng.normal(0.0, 1.0, (Nl, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        dcoef = -np.sum(nrm * lm_pts, axis=1)
        lms_true = np.concatenate([nrm, dcoef[:, None]], axis=1)
        z_true = plane_transform_to_local(poses_true[pi], lms_true[li])
        nn = z_true[:, :3] + rng.normal(0.0, 0.01 * noise_scale, (El, 3))
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        dd = z_true[:, 3] + rng.normal(0.0, 0.05 * noise_scale, El)
        lm_zm = np.concatenate([nn, dd[:, None]], axis=1)
        lms_init = np.zeros((Nl, 4))
        seen = np.zeros(Nl, bool)
        for e in range(El):
            l = li[e]
            if not seen[l]:
                seen[l] = True
                lms_init[l] = plane_transform_to_world(poses_init[pi[e]], lm_zm[e])
        lms_init[~seen] = lms_true[~seen]
    else:
        raise ValueError(landmark_kind)

    return SynthGraph(poses_true, poses_init, lms_true, lms_init, odom_ij, odom_z, odom_info,
                      lm_ij, lm_zm, lm_info, landmark_kind)


GRAPH_S = dict(n_poses=500, n_landmarks=100)     # BASELINE.json configs[1]
GRAPH_L = dict(n_poses=5000, n_landmarks=1000)   # BASELINE.json configs[2]


# ----------------------------------------------------------------------------
# organised depth cloud generator
# ----------------------------------------------------------------------------

CAM_FX = CAM_FY = 525.0
CAM_CX, CAM_CY = 319.5, 239.5
CLASS_CHAIR = 1   # SSLAM_CLASS_CHAIR, whitelisted at point_cloud_segmentation.h:126-130


@dataclasses.dataclass
class SynthFrame:
    cloud: np.ndarray        # uint8 [height*row_step]  sensor_msgs::PointCloud2::data
    width: int
    height: int
    point_step: int
    row_step: int
    offsets: tuple           # (x, y, z) field offsets
    boxes: np.ndarray        # structured: tl_x, tl_y, width, height, class_id, prob
    robot_pose: np.ndarray   # float32 [6] x y z roll pitch yaw
    cam_angle: float         # radians (config camera_angle 33.93 deg)

    def xyz(self) -> np.ndarray:
        a = self.cloud.view(np.float32).reshape(self.height, self.width, self.point_step // 4)
        return a[:, :, :3]


def repack_xyz(frame: "SynthFrame", point_step: int = 12) -> "SynthFrame":
    """The same frame with xyz-only points: 12 bytes (packed floats) or 16 (pcl::PointXYZ's padded layout).  The frontend reads 12 of a
    registered cloud's 32 bytes per point (reference src/planar_segmentation/plane_segmentation.cpp:24-82 copies x, y, z of the crop); a
    caller that converts the sensor message anyway (pcl::fromROSMsg in the reference's callback) can hand over this layout and the
    per-frame PCIe copy shrinks from 9.83 MB to 3.69 MB.  The C-ABI takes any point_step / field offsets."""
    assert point_step in (12, 16)
    xyz = np.ascontiguousarray(frame.xyz(), np.float32)
    out = np.zeros((frame.height, frame.width, point_step // 4), np.float32)
    out[:, :, :3] = xyz
    return dataclasses.replace(frame, cloud=out.reshape(-1).view(np.uint8).copy(), point_step=point_step,
                               row_step=point_step * frame.width, offsets=(0, 4, 8))


BOX_DTYPE = np.dtype([("tl_x", "<i4"), ("tl_y", "<i4"), ("width", "<i4"), ("height", "<i4"), ("class_id", "<i4"), ("prob", "<f4")])


def make_frame(seed: int = 0, width: int = 640, height: int = 480, n_boxes: int = 32, box_w: int = 128, box_h: int = 96,
               n_objects: int = 12, nan_fraction: float = 0.0, noise: float = 1.5e-3, n_holes: int = 10) -> SynthFrame:
    """Piecewise-planar scene (floor + back wall + cuboids) seen by a pinhole camera pitched down
    by 33.93 deg (config/bucket_detector.yaml camera_angle); depth noise N(0,(noise*z^2)^2) with SURVEY §8d's
    1.5e-3 (a structured-light sensor's quadratic range noise); missing depth as a real sensor produces it:
    `n_holes` clustered elliptical drop-outs (specular / absorbing patches, 3-14 px radii, ~1.5 % of the image) plus
    everything beyond 9.5 m, NOT salt-and-pepper pixels -- an isolated NaN collapses PCL's distance-map-limited smoothing
    window around it, so 2 % of salt NaNs silence three quarters of the normals (`nan_fraction` keeps that model
    available for the edge-case tests).  point_step 32 with x@0 y@4 z@8 rgb@16 as depth_image_proc emits."""
    rng = np.random.default_rng(seed)
    pitch = np.deg2rad(33.93)
    # world: x right, y forward, z up.  camera axes in world: x_c = right, z_c = forward pitched down, y_c = down
    zc = np.array([0.0, np.cos(pitch), -np.sin(pitch)])
    xc = np.array([1.0, 0.0, 0.0])
    yc = np.cross(zc, xc)
    Rwc = np.stack([xc, yc, zc], axis=1)
    cam = np.array([0.0, 0.0, 1.3])
    u, v = np.meshgrid(np.arange(width), np.arange(height))
    d_c = np.stack([(u - CAM_CX) / CAM_FX, (v - CAM_CY) / CAM_FY, np.ones_like(u, float)], axis=-1)
    d_w = d_c @ Rwc.T
    t_best = np.full((height, width), np.inf)
    # floor z = 0 and back wall y = 7
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -cam[2] / d_w[..., 2]
        t_best = np.where((t > 0) & (t < t_best), t, t_best)
        t = (7.0 - cam[1]) / d_w[..., 1]
        t_best = np.where((t > 0) & (t < t_best), t, t_best)
        for _ in range(n_objects):
            c = np.array([rng.uniform(-2.5, 2.5), rng.uniform(1.5, 6.0), 0.0])
            sz = np.array([rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.2), rng.uniform(0.3, 1.0)])
            lo = c - np.array([sz[0] / 2, sz[1] / 2, 0.0]); hi = c + np.array([sz[0] / 2, sz[1] / 2, sz[2]])
            t0 = (lo - cam) / d_w; t1 = (hi - cam) / d_w
            tn = np.minimum(t0, t1).max(-1); tf = np.maximum(t0, t1).min(-1)
            hit = (tn <= tf) & (tn > 0)
            t_best = np.where(hit & (tn < t_best), tn, t_best)
    z = t_best.copy()
    z[~np.isfinite(z)] = np.nan
    z = z + rng.normal(0.0, 1.0, z.shape) * noise * z * z
    if nan_fraction > 0:
        z[rng.uniform(size=z.shape) < nan_fraction] = np.nan
    for _ in range(n_holes):   # clustered drop-outs
        cu, cv = rng.uniform(0, width), rng.uniform(0, height)
        ru, rv = rng.uniform(3, 14), rng.uniform(3, 14)
        z[((u - cu) / ru) ** 2 + ((v - cv) / rv) ** 2 <= 1.0] = np.nan
    z[z > 9.5] = np.nan
    pts = np.stack([(u - CAM_CX) / CAM_FX * z, (v - CAM_CY) / CAM_FY * z, z], axis=-1).astype(np.float32)
    point_step = 32
    buf = np.zeros((height, width, point_step // 4), np.float32)
    buf[:, :, :3] = pts
    buf[:, :, 4] = rng.uniform(0, 1, (height, width)).astype(np.float32)  # packed rgb stand-in
    boxes = np.zeros(n_boxes, BOX_DTYPE)
    boxes["tl_x"] = rng.integers(0, width - box_w + 1, n_boxes)
    boxes["tl_y"] = rng.integers(0, height - box_h + 1, n_boxes)
    boxes["width"] = box_w; boxes["height"] = box_h
    boxes["class_id"] = CLASS_CHAIR
    boxes["prob"] = rng.uniform(0.5, 1.0, n_boxes).astype(np.float32)
    pose = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 1.3, 0.0, 0.0, rng.uniform(-np.pi, np.pi)], np.float32)
    return SynthFrame(buf.reshape(-1).view(np.uint8).copy(), width, height, point_step, point_step * width, (0, 4, 8), boxes,
                      pose, float(pitch))


# ----------------------------------------------------------------------------
# recorded-run stand-in for the orchestrator tick
# ----------------------------------------------------------------------------

@dataclasses.dataclass
class ReplayEvent:
    """One odometry sample of a synthetic run: what the node's callbacks would have received before ``VIOCallback``."""
    stamp: tuple                 # (sec, nsec)
    odom: np.ndarray             # [t, q(x,y,z,w)] drifting odometry
    true_pose: np.ndarray        # [t, q] ground truth (not shown to the system)
    objects: list | None         # segmented objects seen from this pose (dicts: pose, normal, class_id, plane_type) or None
    run_after: bool              # the node's loop calls run() after this sample


def make_replay(seed: int = 0, n_samples: int = 400, n_landmarks: int = 24, *, rate_hz: float = 10.0, speed: float = 1.0,
                detect_every: int = 1, detect_from: int = 130, run_every: int = 1, first_run_at: int = 130, view_range: float = 4.0,
                meas_sigma: float = 0.02, odom_sigma_t: float = 0.002, odom_sigma_r: float = 0.0005):
    """A robot driving a closed planar loop past point landmarks of two classes and two plane types, sampled at ``rate_hz``:
    drifting odometry (what ``VIOCallback`` gets, semantic_graph_slam.cpp:234), and every ``detect_every`` samples the objects the
    frontend would have returned for that view (centroid in the camera frame, z forward / x right / y down, the frame
    ``transformNormalsToWorld`` maps from, tools.h:18-102).  ``run()`` is withheld until ``first_run_at`` so that the first tick
    finds more than ``max_keyframes_per_update`` keyframes queued (semantic_graph_slam.cpp:113-116); detections start at
    ``detect_from`` (before the first tick every keyframe carries robot_pose_ = identity, semantic_graph_slam.cpp:46,274-276, and a
    keyframe's own odometry increment never reaches robot_pose_, :239-262 -- with detections inside a stall the association works
    on those stale poses, which the parity tests exercise separately)."""
    rng = np.random.default_rng(seed)
    R = max(speed * n_samples / rate_hz / (2 * np.pi * 1.15), 3.0)   # 1.15 laps: the first landmarks are seen again at the end
    s = np.arange(n_samples) * speed / rate_hz / R
    xy = np.stack([R * np.cos(s) - R, R * np.sin(s)], -1)
    yaw = s + np.pi / 2
    true = np.zeros((n_samples, 7))
    true[:, :2] = xy
    true[:, 5] = np.sin(yaw / 2); true[:, 6] = np.cos(yaw / 2)
    # drifting odometry: noisy increments of the true motion
    odom = np.zeros_like(true); odom[0] = true[0]
    for k in range(1, n_samples):
        inc = pose_compose(pose_inverse(true[k - 1]), true[k])
        inc[:3] += rng.normal(0, odom_sigma_t, 3) * np.array([1, 1, 0])
        inc[3:] = quat_mul(inc[3:], quat_from_rotvec(rng.normal(0, odom_sigma_r, 3) * np.array([0, 0, 1])))
        odom[k] = pose_compose(odom[k - 1], inc)
    # landmarks in a band around the track, at least 2.2 m apart
    lms = []
    tries = 0
    while len(lms) < n_landmarks and tries < 20000:
        tries += 1
        a = rng.uniform(0, 2 * np.pi); r = R + rng.choice([-1, 1]) * rng.uniform(1.0, 2.5)
        p = np.array([r * np.cos(a) - R, r * np.sin(a), rng.uniform(0.2, 1.2)])
        if all(np.linalg.norm(p - q[0]) > 2.2 for q in lms):
            lms.append((p, int(rng.integers(1, 3)), int(rng.integers(0, 2))))
    # camera axes in the robot frame: Rz(-90) Rx(-90) (tools.h:104-135 with zero camera pitch)
    c2r = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    events = []
    for k in range(n_samples):
        objs = None
        if k % detect_every == 0 and k >= detect_from:
            objs = []
            for p, cls, pt in lms:
                pr = quat_rotate(quat_conj(true[k, 3:]), p - true[k, :3])
                d = np.linalg.norm(pr[:2])
                if 0.5 < d < view_range and abs(np.arctan2(pr[1], pr[0])) < np.pi / 3:
                    pc = c2r.T @ pr + rng.normal(0, meas_sigma, 3)
                    nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
                    objs.append(dict(pose=pc.astype(np.float32), normal=np.append(nrm, rng.uniform(-2, 2)).astype(np.float32),
                                     class_id=cls, plane_type=pt))
        t = k / rate_hz
        sec = int(np.floor(t + 1e-9)); nsec = int(round((t - sec) * 1e9))
        events.append(ReplayEvent((sec, nsec), odom[k], true[k], objs, k >= first_run_at and (k - first_run_at) % run_every == 0))
    events[-1].run_after = True
    return events, lms


def _main(argv):
    """python -m semantic_slam_amd.synth KIND POSES LANDMARKS OUT_DIR SEED...  -> OUT_DIR/KIND_POSES_LANDMARKS_SEED.g2o
    (bench.py's workload generator: plain host processes, no HIP call is made)"""
    import os
    from .graph_slam import GraphSLAM
    kind, poses, landmarks, out_dir = argv[0], int(argv[1]), int(argv[2]), argv[3]
    for sd in argv[4:]:
        path = os.path.join(out_dir, f"{kind}_{poses}_{landmarks}_{int(sd)}.g2o")
        if os.path.exists(path) and os.path.getsize(path) > 0:
            continue
        G = GraphSLAM.from_synth(make_graph(poses, landmarks, seed=int(sd), landmark_kind=kind))
        tmp = f"{path}.{os.getpid()}.tmp"
        G.save(tmp)
        os.replace(tmp, path)


if __name__ == "__main__":
    import sys
    _main(sys.argv[1:])
