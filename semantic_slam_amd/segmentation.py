"""Host-side mirror of ``point_cloud_segmentation`` over the C-ABI (include/sslam.h).

``PointCloudSegmentation.segmentallPointCloudData(robot_pose, cam_angle, object_info, point_cloud)``
has the argument meaning of the reference entry point
(reference include/planar_segmentation/point_cloud_segmentation.h:105-108); detections come back as
``DetectedObject`` records with the fields of reference include/planar_segmentation/detected_object.h:14-24.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Sequence

import numpy as np

from ._lib import load_library
from . import synth

CLASS_NAMES = ["other", "chair", "tvmonitor", "book", "keyboard", "laptop", "bucket", "car"]  # point_cloud_segmentation.h:126-130


class SegParams(C.Structure):
    _fields_ = [("num_point_seg", C.c_double), ("norm_point_thres", C.c_double), ("planar_area", C.c_double),
                ("max_depth_change_factor", C.c_float), ("normal_smoothing_size", C.c_float), ("angular_threshold", C.c_float),
                ("distance_threshold", C.c_float), ("maximum_curvature", C.c_float), ("min_contour_points", C.c_int),
                ("image_width", C.c_int), ("image_height", C.c_int), ("reference_quirks", C.c_int), ("device", C.c_int)]


class Box(C.Structure):
    _fields_ = [("tl_x", C.c_int32), ("tl_y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("class_id", C.c_int32), ("prob", C.c_float)]


class Plane(C.Structure):
    _fields_ = [("centroid_cam", C.c_float * 3), ("normal_d", C.c_float * 4), ("world_pose", C.c_float * 3),
                ("num_points", C.c_float), ("prob", C.c_float), ("plane_type", C.c_int32), ("class_id", C.c_int32),
                ("box_index", C.c_int32), ("inlier_count", C.c_int32), ("area", C.c_float)]


class Frame(C.Structure):   # sslam_frame
    _fields_ = [("cloud", C.c_void_p), ("boxes", C.c_void_p), ("n_boxes", C.c_int), ("robot_pose", C.c_float * 6), ("cam_angle", C.c_float)]


_PLANE_DTYPE = np.dtype([("centroid_cam", "<f4", (3,)), ("normal_d", "<f4", (4,)), ("world_pose", "<f4", (3,)), ("num_points", "<f4"), ("prob", "<f4"),
                         ("plane_type", "<i4"), ("class_id", "<i4"), ("box_index", "<i4"), ("inlier_count", "<i4"), ("area", "<f4")])   # = Plane


@dataclasses.dataclass
class DetectedObject:
    """detected_object.h:14-24"""
    prob: float
    num_points: float
    type: str
    plane_type: str
    pose: np.ndarray
    world_pose: np.ndarray
    normal_orientation: np.ndarray
    box_index: int = -1
    inlier_count: int = 0
    area: float = 0.0


_BOUND = False


class BoxPlane(C.Structure):
    """``sslam_box_plane`` (include/sslam.h)"""
    _fields_ = [("coeff", C.c_float * 4), ("inliers", C.c_int32), ("points", C.c_int32), ("box_index", C.c_int32), ("frame", C.c_int32),
                ("hypotheses", C.c_int32), ("best_iteration", C.c_int32)]


class IcpResult(C.Structure):
    """``sslam_icp_result`` (include/sslam.h)"""
    _fields_ = [("T", C.c_double * 12), ("rms", C.c_double), ("used", C.c_int32), ("status", C.c_int32)]


def _bind(lib):
    global _BOUND
    if _BOUND:
        return
    vp, ci = C.c_void_p, C.c_int
    lib.sslam_seg_default_params.restype = None; lib.sslam_seg_default_params.argtypes = [C.POINTER(SegParams)]
    lib.sslam_seg_create.restype = vp; lib.sslam_seg_create.argtypes = [C.POINTER(SegParams)]
    lib.sslam_seg_destroy.restype = None; lib.sslam_seg_destroy.argtypes = [vp]
    lib.sslam_seg_segment.restype = ci
    lib.sslam_seg_segment.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp, C.c_float, vp, ci]
    lib.sslam_seg_segment_batch.restype = ci
    lib.sslam_seg_segment_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp]
    lib.sslam_seg_submit_batch.restype = ci
    lib.sslam_seg_submit_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci]
    lib.sslam_seg_collect_batch.restype = ci
    lib.sslam_seg_collect_batch.argtypes = [vp, vp, ci, vp]
    lib.sslam_seg_last_overflow.restype = ci; lib.sslam_seg_last_overflow.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.sslam_seg_get_normals.restype = ci; lib.sslam_seg_get_normals.argtypes = [vp, ci, vp]
    lib.sslam_seg_get_labels.restype = ci; lib.sslam_seg_get_labels.argtypes = [vp, ci, vp]
    lib.sslam_seg_last_timing.restype = ci; lib.sslam_seg_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.sslam_seg_transform.restype = ci; lib.sslam_seg_transform.argtypes = [vp, vp, C.c_float, vp]
    lib.sslam_seg_ransac_plane.restype = ci
    lib.sslam_seg_ransac_plane.argtypes = [vp, vp, ci, C.c_float, ci, C.c_double, C.c_uint64, vp, vp, ci]
    lib.sslam_seg_convex_hull_2d.restype = ci
    lib.sslam_seg_convex_hull_2d.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, ci, vp]
    lib.sslam_seg_distance_filter.restype = ci; lib.sslam_seg_distance_filter.argtypes = [vp, vp, ci, C.c_double, C.c_double, vp, ci]
    lib.sslam_seg_voxel_grid.restype = ci; lib.sslam_seg_voxel_grid.argtypes = [vp, vp, ci, C.c_float, vp, vp, ci]
    lib.sslam_seg_statistical_outlier_removal.restype = ci
    lib.sslam_seg_statistical_outlier_removal.argtypes = [vp, vp, ci, ci, C.c_double, vp, ci, vp]
    lib.sslam_seg_kmeans.restype = ci; lib.sslam_seg_kmeans.argtypes = [vp, vp, ci, ci, ci, C.c_uint64, vp, vp, C.POINTER(C.c_double)]
    lib.sslam_seg_ransac_boxes.restype = ci
    lib.sslam_seg_ransac_boxes.argtypes = [vp, C.c_float, ci, C.c_double, C.c_uint64, vp, ci, C.POINTER(C.c_double)]
    lib.sslam_seg_ransac_box_inliers.restype = ci
    lib.sslam_seg_ransac_box_inliers.argtypes = [vp, ci, vp, ci]
    lib.sslam_seg_icp_boxes.restype = ci
    lib.sslam_seg_icp_boxes.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, ci, C.POINTER(C.c_double)]
    lib.sslam_seg_icp_point_to_plane.restype = ci
    lib.sslam_seg_icp_point_to_plane.argtypes = [vp, vp, vp, ci, vp, ci, ci, vp, vp, C.POINTER(C.c_double)]
    _BOUND = True


def default_params(device: int = 0) -> SegParams:
    lib = load_library(); _bind(lib)
    p = SegParams()
    lib.sslam_seg_default_params(C.byref(p))
    p.device = device
    return p


class PointCloudSegmentation:
    """``point_cloud_segmentation`` + ``plane_segmentation`` on one MI355X."""

    def __init__(self, verbose: bool = False, params: SegParams | None = None, device: int = 0):
        self._lib = load_library(); _bind(self._lib)
        self.verbose_ = verbose
        self.params = params if params is not None else default_params(device)
        self._h = self._lib.sslam_seg_create(C.byref(self.params))
        self._last_boxes = None
        self._inflight = []          # (frame count, keep-alive buffers) of every submitted, not yet collected batch

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sslam_seg_destroy(h)

    def _check(self, rc):
        if rc < 0:
            from .graph_slam import SslamError
            raise SslamError(rc, self._lib.sslam_last_error().decode())
        return rc

    def segmentallPointCloudData(self, robot_pose, cam_angle: float, object_info, point_cloud, *, width=None, height=None,
                                 point_step=None, row_step=None, offsets=(0, 4, 8), max_planes: int = 512) -> List[DetectedObject]:
        """point_cloud_segmentation.h:105-181.  ``object_info``: structured array / sequence of
        (tl_x, tl_y, width, height, class_id | class name, prob).  ``point_cloud``: a
        ``synth.SynthFrame``-like object (cloud bytes + layout) or raw ``uint8`` bytes with the layout given."""
        if hasattr(point_cloud, "cloud"):
            f = point_cloud
            cloud, width, height, point_step, row_step, offsets = f.cloud, f.width, f.height, f.point_step, f.row_step, f.offsets
        else:
            cloud = np.ascontiguousarray(point_cloud, np.uint8)
        boxes = self._boxes(object_info)
        pose = np.ascontiguousarray(robot_pose, np.float32).reshape(6)
        out = (Plane * max_planes)()
        n = self._check(self._lib.sslam_seg_segment(self._h, cloud.ctypes.data, width, height, point_step, row_step,
                                                    offsets[0], offsets[1], offsets[2], C.cast(boxes, C.c_void_p), len(boxes),
                                                    pose.ctypes.data, C.c_float(cam_angle), C.cast(out, C.c_void_p), max_planes))
        self._last_boxes = boxes
        return self._objects(out, n)

    @staticmethod
    def _boxes(object_info):
        # a structured array in the C layout (synth.BOX_DTYPE: four int32, class id, float probability) is copied as it stands -- per-box
        # Python conversions cost 3 ms per batch of 32 x 32 boxes, more than the GPU needs for the batch
        # (the dtype itself must match -- names, field types AND offsets: an array with class_id as float32 has the same item size and would be
        # reinterpreted silently; it takes the converting path below instead)
        if isinstance(object_info, np.ndarray) and object_info.dtype == synth.BOX_DTYPE and object_info.dtype.itemsize == C.sizeof(Box):
            return (Box * len(object_info)).from_buffer_copy(np.ascontiguousarray(object_info).tobytes()) if len(object_info) else (Box * 0)()
        boxes = (Box * len(object_info))()
        for k, o in enumerate(object_info):
            cls = o[4]
            if isinstance(cls, (str, bytes)):
                name = cls.decode() if isinstance(cls, bytes) else cls
                cls = CLASS_NAMES.index(name) if name in CLASS_NAMES else 0
            boxes[k] = Box(int(o[0]), int(o[1]), int(o[2]), int(o[3]), int(cls), float(o[5]))
        return boxes

    @staticmethod
    def _objects(out, n):
        # one view of the C records instead of ~15 ctypes attribute reads per plane (0.5 -> 0.1 ms per batch of 32 frames: in the pipelined
        # calls the host has to hand the next batch over before the running one ends)
        if n <= 0:
            return []
        rec = np.frombuffer(out, dtype=_PLANE_DTYPE, count=n).copy()
        cc, nd, wp = rec["centroid_cam"], rec["normal_d"], rec["world_pose"]
        prob, npts, pt, cls = rec["prob"].tolist(), rec["num_points"].tolist(), rec["plane_type"].tolist(), rec["class_id"].tolist()
        bi, ic, ar = rec["box_index"].tolist(), rec["inlier_count"].tolist(), rec["area"].tolist()
        return [DetectedObject(prob=prob[k], num_points=npts[k], type=CLASS_NAMES[cls[k]], plane_type="horizontal" if pt[k] == 0 else "vertical",
                               pose=cc[k], world_pose=wp[k], normal_orientation=nd[k], box_index=bi[k], inlier_count=ic[k], area=ar[k])
                for k in range(n)]

    def _pack_frames(self, frames):
        F = len(frames)
        f0 = frames[0]
        fr = (Frame * F)()
        keep = []
        for k, f in enumerate(frames):
            assert (f.width, f.height, f.point_step, f.row_step, tuple(f.offsets)) == (f0.width, f0.height, f0.point_step, f0.row_step, tuple(f0.offsets))
            boxes = self._boxes(f.boxes)
            pose = np.ascontiguousarray(f.robot_pose, np.float32).reshape(6)
            keep.append((boxes, f.cloud))
            fr[k].cloud = f.cloud.ctypes.data; fr[k].boxes = C.cast(boxes, C.c_void_p); fr[k].n_boxes = len(boxes)
            for q in range(6):
                fr[k].robot_pose[q] = float(pose[q])
            fr[k].cam_angle = float(f.cam_angle)
        return fr, keep

    def submit_frames(self, frames) -> None:
        """``sslam_seg_submit_batch``: enqueue one batch of frames (H2D copy + kernels) and return; at most two batches in flight.
        The frames' clouds are kept alive until the batch is collected."""
        fr, keep = self._pack_frames(frames)
        f0 = frames[0]
        self._check(self._lib.sslam_seg_submit_batch(self._h, C.cast(fr, C.c_void_p), len(frames), f0.width, f0.height, f0.point_step, f0.row_step,
                                                     f0.offsets[0], f0.offsets[1], f0.offsets[2]))
        self._inflight.append((len(frames), keep))

    def collect_frames(self, max_planes: int = 4096):
        """``sslam_seg_collect_batch``: the planes of the oldest submitted batch, per frame (as ``segment_frames`` returns them)"""
        out = (Plane * max_planes)()
        which = np.zeros(max_planes, np.int32)
        # the C side consumes the oldest slot of its fifo whether or not the collect succeeds: the keep-alive list follows it in step
        # (nothing in flight: the C call refuses and _check raises)
        F, keep = self._inflight.pop(0) if self._inflight else (0, [])
        n = self._check(self._lib.sslam_seg_collect_batch(self._h, C.cast(out, C.c_void_p), max_planes, which.ctypes.data))
        self._last_keep = keep                                           # accepted slot = position, when no box is rejected (read lazily)
        self._last_boxes = None
        res = [[] for _ in range(F)]
        for k, o in enumerate(self._objects(out, n)):
            res[int(which[k])].append(o)
        return res

    def segment_stream(self, batches, max_planes: int = 4096):
        """generator over an iterable of frame batches: batch k+1 is submitted before batch k is collected, so its H2D copy runs under
        the kernels of batch k"""
        it = iter(batches)
        try:
            self.submit_frames(next(it))
        except StopIteration:
            return
        for nxt in it:
            self.submit_frames(nxt)
            yield self.collect_frames(max_planes)
        yield self.collect_frames(max_planes)

    def segment_frames(self, frames, max_planes: int = 4096):
        """Several frames (``synth.SynthFrame``-like objects with robot_pose / cam_angle / boxes) in ONE pass over the GPU
        (``sslam_seg_segment_batch``): the boxes of all frames share every kernel launch.  Returns a list (per frame) of lists of
        planes, identical to calling ``segmentallPointCloudData`` frame by frame."""
        F = len(frames)
        if F == 0:
            return []
        f0 = frames[0]
        fr, keep = self._pack_frames(frames)
        out = (Plane * max_planes)()
        which = np.zeros(max_planes, np.int32)
        n = self._check(self._lib.sslam_seg_segment_batch(self._h, C.cast(fr, C.c_void_p), F, f0.width, f0.height, f0.point_step, f0.row_step,
                                                          f0.offsets[0], f0.offsets[1], f0.offsets[2], C.cast(out, C.c_void_p), max_planes,
                                                          which.ctypes.data))
        self._last_keep = keep                                           # accepted slot = position, when no box is rejected (read lazily)
        self._last_boxes = None
        objs = self._objects(out, n)
        res = [[] for _ in range(F)]
        for k, o in enumerate(objs):
            res[int(which[k])].append(o)
        return res

    def last_overflow(self):
        """(planes dropped because max_planes was reached, boxes with a full candidate table, boxes with a full region table)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._lib.sslam_seg_last_overflow(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def ransac_plane(self, xyz, threshold: float = 0.01, max_iterations: int = 50, probability: float = 0.99, seed: int = 0):
        """pcl::SACSegmentation plane RANSAC of plane_segmentation::compute2DConvexHull (plane_segmentation.cpp:639-647).
        Returns (coefficients[4], ascending inlier indices)."""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        coeff = np.zeros(4, np.float32)
        inl = np.zeros(max(len(pts), 1), np.int32)
        n = self._check(self._lib.sslam_seg_ransac_plane(self._h, pts.ctypes.data, len(pts), C.c_float(threshold), max_iterations,
                                                          C.c_double(probability), C.c_uint64(seed), coeff.ctypes.data, inl.ctypes.data, len(inl)))
        return coeff, inl[:n].copy()

    def ransac_boxes(self, threshold: float = 0.01, max_iterations: int = 50, probability: float = 0.99, seed: int = 0, max_boxes: int = 4096):
        """``sslam_seg_ransac_boxes``: RANSAC plane of every accepted box of the resident batch (the last ``segment_frames`` /
        ``segmentallPointCloudData`` call), one workgroup per box.  Returns (list of BoxPlane records in slot order, kernel ms)."""
        out = (BoxPlane * max_boxes)()
        ms = C.c_double(0)
        n = self._check(self._lib.sslam_seg_ransac_boxes(self._h, C.c_float(threshold), max_iterations, C.c_double(probability), C.c_uint64(seed),
                                                          C.cast(out, C.c_void_p), max_boxes, C.byref(ms)))
        return [out[k] for k in range(min(n, max_boxes))], ms.value

    def ransac_box_inliers(self, slot: int, max_points: int = 640 * 480):
        """ascending crop indices (row-major inside the box) of the inliers of box `slot`'s refined model"""
        buf = np.zeros(max_points, np.int32)
        n = self._check(self._lib.sslam_seg_ransac_box_inliers(self._h, slot, buf.ctypes.data, max_points))
        return buf[:min(n, max_points)].copy()

    def icp_boxes(self, box_plane, planes, iterations: int = 10, T0=None, max_frames: int = 1024):
        """``sslam_seg_icp_boxes``: point-to-plane ICP per frame of the resident batch over the RANSAC inliers of its boxes; box slot q
        measures plane box_plane[q] (-1: none).  Returns (list of IcpResult per frame, kernel ms)."""
        bp = np.ascontiguousarray(box_plane, np.int32)
        pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
        t0 = None if T0 is None else np.ascontiguousarray(T0, np.float64).reshape(-1)
        out = (IcpResult * max_frames)()
        ms = C.c_double(0)
        n = self._check(self._lib.sslam_seg_icp_boxes(self._h, bp.ctypes.data, len(bp), pl.ctypes.data, len(pl), iterations,
                                                       None if t0 is None else t0.ctypes.data, C.cast(out, C.c_void_p), max_frames, C.byref(ms)))
        return [out[k] for k in range(min(n, max_frames))], ms.value

    def convex_hull_2d(self, xyz, inliers, coeff):
        """pcl::ProjectInliers + 2-D pcl::ConvexHull of plane_segmentation::compute2DConvexHull (plane_segmentation.cpp:648-662).
        Returns (projected[n_inliers, 3], hull positions into `inliers` in angular order, axes)."""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        inl = np.ascontiguousarray(inliers, np.int32)
        co = np.ascontiguousarray(coeff, np.float32).reshape(4)
        proj = np.zeros((len(inl), 3), np.float32)
        hull = np.zeros(max(len(inl), 1), np.int32)
        axes = C.c_int(-1)
        h = self._check(self._lib.sslam_seg_convex_hull_2d(self._h, pts.ctypes.data, len(pts), inl.ctypes.data, len(inl), co.ctypes.data,
                                                            proj.ctypes.data, hull.ctypes.data, len(hull), C.byref(axes)))
        return proj, hull[:h].copy(), axes.value


    # ---- cloud filters of the legacy path --------------------------------------------------------------------
    def distance_filter(self, xyz, dmin: float = 0.3, dmax: float = 3.0):
        """plane_segmentation::distance_filter (plane_segmentation.cpp:607-629): ascending indices of the points kept"""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        out = np.zeros(max(len(pts), 1), np.int32)
        n = self._check(self._lib.sslam_seg_distance_filter(self._h, pts.ctypes.data, len(pts), dmin, dmax, out.ctypes.data, len(pts)))
        return out[:n]

    def downsamplePointcloud(self, xyz, leaf: float = 0.1):
        """pcl::VoxelGrid (plane_segmentation.cpp:565-581): (centroids [m, 3], points per voxel [m]) in ascending voxel index"""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        cap = max(len(pts), 1)
        cent = np.zeros((cap, 3), np.float32); cnt = np.zeros(cap, np.int32)
        m = self._check(self._lib.sslam_seg_voxel_grid(self._h, pts.ctypes.data, len(pts), C.c_float(leaf), cent.ctypes.data, cnt.ctypes.data, cap))
        return cent[:m], cnt[:m]

    def removeOutliers(self, xyz, mean_k: int = 50, stddev_mul: float = 1.0):
        """pcl::StatisticalOutlierRemoval (plane_segmentation.cpp:583-605): (ascending inlier indices, mean neighbour distance per point)"""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        out = np.zeros(max(len(pts), 1), np.int32); md = np.zeros(max(len(pts), 1), np.float32)
        n = self._check(self._lib.sslam_seg_statistical_outlier_removal(self._h, pts.ctypes.data, len(pts), mean_k, stddev_mul, out.ctypes.data,
                                                                        len(pts), md.ctypes.data))
        return out[:n], md[:len(pts)]

    def computeKmeans(self, points, num_centroids: int, seed: int = 0):
        """plane_segmentation::computeKmeans (plane_segmentation.cpp:524-535) -> (labels, centroids, compactness)"""
        pts = np.ascontiguousarray(points, np.float32)
        pts = pts.reshape(len(pts), -1)
        n, dim = pts.shape
        lab = np.zeros(n, np.int32); cen = np.zeros((num_centroids, dim), np.float32); comp = C.c_double(0)
        self._check(self._lib.sslam_seg_kmeans(self._h, pts.ctypes.data, n, dim, num_centroids, seed, lab.ctypes.data, cen.ctypes.data, C.byref(comp)))
        return lab, cen, comp.value

    def clusterAndSegmentAllPlanes(self, xyz, normals, transformation_mat, seed: int = 0, kmeans=None, hull=None):
        """plane_segmentation::clusterAndSegmentAllPlanes (plane_segmentation.cpp:261-497; dead code upstream): k-means on the normals
        (4 centres) -> centres within +-0.3 of the horizontal-plane normal seen from the camera -> per centre a k-means on the plane
        distances (2 centres) -> clusters of more than 500 points -> compute2DConvexHull.  Returns the 1 x 8 rows the reference
        builds, [hull x, y, z, normal x, y, z, distance, 0].  `kmeans` / `hull` replace the GPU primitives (the tests pass the oracle's)."""
        kmeans = kmeans or self.computeKmeans
        hull = hull or self.compute2DConvexHull
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        ok = ~np.isnan(nrm).any(1)                                     # removeNans (:479-497)
        P, N = pts[ok], nrm[ok]
        if len(N) <= 10:
            return np.zeros((0, 8), np.float32)
        labels, centers, _ = kmeans(N, 4, seed)                        # num_centroids_normals (plane_segmentation.h:41)
        T = np.asarray(transformation_mat, np.float32).reshape(4, 4)
        n_cam = T.T @ np.array([0, 0, 1, 0], np.float32)               # (:343-345)
        rows = []
        for cid in range(len(centers)):
            c = centers[cid]
            if not all(n_cam[d] - 0.3 < c[d] < n_cam[d] + 0.3 for d in range(3)):     # filterCentroids (:499-518)
                continue
            Pi = P[labels == cid]
            if len(Pi) < 2:
                continue
            dist = -((Pi[:, 0] * c[0] + Pi[:, 1] * c[1]).astype(np.float32) + Pi[:, 2] * c[2]).astype(np.float32)   # (:383-391)
            dl, dc, _ = kmeans(dist.reshape(-1, 1), 2, seed + 1 + cid)     # num_centroids_distance (plane_segmentation.h:42)
            for d in range(2):
                Q = Pi[dl == d]
                if len(Q) > 500:                                           # (:418)
                    for h in hull(Q, seed):
                        rows.append([h[0], h[1], h[2], c[0], c[1], c[2], dc[d, 0], 0.0])
        return np.asarray(rows, np.float32).reshape(-1, 8)

    def icp_point_to_plane(self, xyz, labels, planes, iterations: int = 10, T0=None):
        """Point-to-plane ICP of labelled points against planes (row J1, ``sslam_seg_icp_point_to_plane``).
        Returns (T[12] = R row-major | t, rms, points used)."""
        pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        lab = np.ascontiguousarray(labels, np.int32).reshape(-1)
        pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
        t0 = None if T0 is None else np.ascontiguousarray(T0, np.float64).reshape(12)
        out = np.zeros(12); rms = C.c_double(0.0)
        n = self._check(self._lib.sslam_seg_icp_point_to_plane(self._h, pts.ctypes.data, lab.ctypes.data, len(pts), pl.ctypes.data, len(pl),
                                                               int(iterations), None if t0 is None else t0.ctypes.data, out.ctypes.data,
                                                               C.byref(rms)))
        return out, rms.value, n

    def compute2DConvexHull(self, xyz, seed: int = 0):
        """plane_segmentation::compute2DConvexHull (plane_segmentation.cpp:631-665): RANSAC plane (threshold 0.01, refined
        coefficients) -> project the inliers -> 2-D convex hull.  Returns the hull points (h x 3, projected coordinates)."""
        coeff, inl = self.ransac_plane(xyz, 0.01, 50, 0.99, seed)
        if len(inl) < 3:                      # no model: an empty hull, like the C++ shim
            return np.zeros((0, 3), np.float32)
        proj, hull, _ = self.convex_hull_2d(xyz, inl, coeff)
        return proj[hull]

    # parity hooks -----------------------------------------------------------------------------
    def normals(self, box: int) -> np.ndarray:
        if self._last_boxes is None and getattr(self, "_last_keep", None) is not None:
            self._last_boxes = [bx for (boxes, _) in self._last_keep for bx in boxes]
        b = self._last_boxes[box]
        out = np.zeros((b.height, b.width, 4), np.float32)
        self._check(self._lib.sslam_seg_get_normals(self._h, box, out.ctypes.data))
        return out

    def labels(self, box: int) -> np.ndarray:
        if self._last_boxes is None and getattr(self, "_last_keep", None) is not None:
            self._last_boxes = [bx for (boxes, _) in self._last_keep for bx in boxes]
        b = self._last_boxes[box]
        out = np.zeros((b.height, b.width), np.int32)
        self._check(self._lib.sslam_seg_get_labels(self._h, box, out.ctypes.data))
        return out

    def last_timing(self):
        k = C.c_double(0); t = C.c_double(0)
        self._lib.sslam_seg_last_timing(self._h, C.byref(k), C.byref(t))
        return k.value, t.value

    def transform(self, robot_pose, cam_angle: float) -> np.ndarray:
        pose = np.ascontiguousarray(robot_pose, np.float32).reshape(6)
        out = np.zeros(16, np.float32)
        self._check(self._lib.sslam_seg_transform(self._h, pose.ctypes.data, C.c_float(cam_angle), out.ctypes.data))
        return out.reshape(4, 4)
