"""Host-side mirror of the reference's orchestrator, ``semantic_graph_slam`` (reference include/ps_graph_slam/semantic_graph_slam.h,
src/ps_graph_slam/semantic_graph_slam.cpp), over the C-ABI ``sslam_slam_*`` of include/sslam.h.

Method names follow the reference (``VIOCallback``, ``run``, ``setPointCloudData``, ``setDetectedObjectInfo``, ``getRobotPose``,
``getMap2OdomTrans``, ``getMappedLandmarks``, ``getKeyframes``, ``saveGraph``); the ROS parameters become ``SlamParams`` fields.
There is no CPU fallback: the tick's frontend pass, optimisation, marginals and data association all run on the GPU.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from ._lib import load_library, OptStats
from .graph_slam import SslamError, _pose7
from .segmentation import Plane, PointCloudSegmentation


class SlamParams(C.Structure):   # sslam_slam_params
    _fields_ = [("keyframe_delta_trans", C.c_double), ("keyframe_delta_angle", C.c_double), ("keyframe_delta_time", C.c_double),
                ("max_keyframes_per_update", C.c_int), ("update_keyframes_using_detections", C.c_int),
                ("camera_angle_deg", C.c_double), ("add_first_lan", C.c_int), ("first_lan", C.c_double * 3),
                ("use_const_inf_matrix", C.c_int), ("const_stddev_x", C.c_double), ("const_stddev_q", C.c_double),
                ("maha_dist_thres", C.c_double), ("eq_dist_thres", C.c_double), ("land_noise_low", C.c_double),
                ("land_noise_high", C.c_double), ("use_maha_dist", C.c_int), ("use_eq_dist", C.c_int), ("use_rtab_map_odom", C.c_int),
                ("max_iterations", C.c_int), ("reference_quirks", C.c_int), ("device", C.c_int)]


class Landmark(C.Structure):     # sslam_landmark (include/ps_graph_slam/landmark.h:17-33)
    _fields_ = [("id", C.c_int32), ("vertex", C.c_int32), ("class_id", C.c_int32), ("plane_type", C.c_int32), ("is_new", C.c_int32),
                ("pose", C.c_float * 3), ("local_pose", C.c_float * 3), ("covariance", C.c_float * 9), ("normal", C.c_float * 4),
                ("distance", C.c_float)]


class TickStats(C.Structure):    # sslam_tick_stats
    _fields_ = [("keyframes_added", C.c_int), ("landmarks_added", C.c_int), ("landmarks_matched", C.c_int),
                ("landmark_edges_added", C.c_int), ("optimized", C.c_int), ("marginals_ok", C.c_int), ("opt", OptStats),
                ("seconds_frontend", C.c_double), ("seconds_association", C.c_double), ("seconds_optimize", C.c_double),
                ("seconds_marginals", C.c_double)]


_BOUND = False


def _bind(lib):
    global _BOUND
    if _BOUND:
        return
    vp, ci = C.c_void_p, C.c_int
    lib.sslam_slam_default_params.restype = None; lib.sslam_slam_default_params.argtypes = [C.POINTER(SlamParams)]
    lib.sslam_slam_create.restype = vp; lib.sslam_slam_create.argtypes = [C.POINTER(SlamParams), vp]
    lib.sslam_slam_destroy.restype = None; lib.sslam_slam_destroy.argtypes = [vp]
    lib.sslam_slam_set_point_cloud.restype = ci; lib.sslam_slam_set_point_cloud.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci]
    lib.sslam_slam_set_detected_objects.restype = ci; lib.sslam_slam_set_detected_objects.argtypes = [vp, vp, ci]
    lib.sslam_slam_set_segmented_objects.restype = ci; lib.sslam_slam_set_segmented_objects.argtypes = [vp, vp, ci]
    lib.sslam_slam_vio.restype = ci; lib.sslam_slam_vio.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.sslam_slam_run.restype = ci; lib.sslam_slam_run.argtypes = [vp, C.POINTER(TickStats)]
    lib.sslam_slam_robot_pose.restype = ci; lib.sslam_slam_robot_pose.argtypes = [vp, vp]
    lib.sslam_slam_map2odom.restype = ci; lib.sslam_slam_map2odom.argtypes = [vp, vp]
    lib.sslam_slam_landmarks.restype = ci; lib.sslam_slam_landmarks.argtypes = [vp, vp, ci]
    lib.sslam_slam_keyframes.restype = ci; lib.sslam_slam_keyframes.argtypes = [vp, vp, vp, ci]
    lib.sslam_slam_graph.restype = vp; lib.sslam_slam_graph.argtypes = [vp]
    lib.sslam_slam_find_matches.restype = ci; lib.sslam_slam_find_matches.argtypes = [vp, vp, ci, vp, vp]
    _BOUND = True


def default_slam_params(device: int = 0) -> SlamParams:
    lib = load_library(); _bind(lib)
    p = SlamParams()
    lib.sslam_slam_default_params(C.byref(p))
    p.device = device
    return p


class SemanticGraphSLAM:
    """``semantic_graph_slam`` without ROS: feed it what the node's callbacks receive, call :meth:`run` where the node's loop does."""

    def __init__(self, params: SlamParams | None = None, segmentation: PointCloudSegmentation | None = None, device: int = 0):
        self._lib = load_library(); _bind(self._lib)
        self.params = params if params is not None else default_slam_params(device)
        self._seg = segmentation   # kept alive: the handle is borrowed by the orchestrator
        self._h = self._lib.sslam_slam_create(C.byref(self.params), segmentation._h if segmentation is not None else None)
        if not self._h:
            raise SslamError(-1, self._lib.sslam_last_error().decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sslam_slam_destroy(h)

    def _check(self, rc):
        if rc < 0:
            raise SslamError(rc, self._lib.sslam_last_error().decode())
        return rc

    # -- callbacks ------------------------------------------------------------------------------------------------------------
    def setPointCloudData(self, point_cloud):
        """semantic_graph_slam.cpp:341-345; ``point_cloud``: object with cloud / width / height / point_step / row_step / offsets"""
        f = point_cloud
        cloud = np.ascontiguousarray(f.cloud, np.uint8)
        self._check(self._lib.sslam_slam_set_point_cloud(self._h, cloud.ctypes.data, f.width, f.height, f.point_step, f.row_step,
                                                         f.offsets[0], f.offsets[1], f.offsets[2]))

    def setDetectedObjectInfo(self, object_info):
        """semantic_graph_slam.cpp:353-357"""
        boxes = PointCloudSegmentation._boxes(object_info)
        self._check(self._lib.sslam_slam_set_detected_objects(self._h, C.cast(boxes, C.c_void_p), len(boxes)))

    def setSegmentedObjects(self, planes):
        """extension: objects already segmented (a list of ``segmentation.Plane`` records) for the next keyframe"""
        arr = (Plane * max(len(planes), 1))(*planes)
        self._check(self._lib.sslam_slam_set_segmented_objects(self._h, C.cast(arr, C.c_void_p), len(planes)))

    def VIOCallback(self, stamp, odom) -> bool:
        """semantic_graph_slam.cpp:234-287.  ``stamp`` = (sec, nsec) or float seconds; ``odom`` = [t, q(x,y,z,w)] or a 4x4 isometry."""
        if isinstance(stamp, (tuple, list)):
            sec, nsec = int(stamp[0]), int(stamp[1])
        else:
            sec = int(np.floor(stamp)); nsec = int(round((stamp - sec) * 1e9))
        tq = _pose7(odom)
        return bool(self._check(self._lib.sslam_slam_vio(self._h, sec, nsec, tq.ctypes.data)))

    # -- the tick -------------------------------------------------------------------------------------------------------------
    def run(self) -> bool:
        """semantic_graph_slam.cpp:58-102; statistics of the tick in ``last_stats``"""
        st = TickStats()
        rc = self._check(self._lib.sslam_slam_run(self._h, C.byref(st)))
        self.last_stats = st
        return bool(rc)

    # -- getters --------------------------------------------------------------------------------------------------------------
    def getRobotPose(self) -> np.ndarray:
        out = np.zeros(7)
        self._check(self._lib.sslam_slam_robot_pose(self._h, out.ctypes.data))
        return out

    def getMap2OdomTrans(self) -> np.ndarray:
        out = np.zeros(7)
        self._check(self._lib.sslam_slam_map2odom(self._h, out.ctypes.data))
        return out

    def getMappedLandmarks(self):
        n = self._check(self._lib.sslam_slam_landmarks(self._h, None, 0))
        arr = (Landmark * max(n, 1))()
        self._check(self._lib.sslam_slam_landmarks(self._h, C.cast(arr, C.c_void_p), n))
        return [arr[k] for k in range(n)]

    def getKeyframes(self):
        """(vertex ids, estimates[n, 7]) of the keyframes already in the graph"""
        n = self._check(self._lib.sslam_slam_keyframes(self._h, None, None, 0))
        ids = np.zeros(max(n, 1), np.int32); est = np.zeros((max(n, 1), 7))
        self._check(self._lib.sslam_slam_keyframes(self._h, ids.ctypes.data, est.ctypes.data, n))
        return ids[:n], est[:n]

    def saveGraph(self, path: str):
        g = self._lib.sslam_slam_graph(self._h)
        rc = self._lib.sslam_graph_save_g2o(g, path.encode())
        if rc < 0:
            raise SslamError(rc, self._lib.sslam_last_error().decode())

    def set_graph_option(self, key: str, value: float) -> None:
        """an option of the orchestrator's graph handle (sslam_graph_set_option through sslam_slam_graph): e.g. "speculative_trials" 0 for
        several robots sharing one GPU (ten speculative lanes fill the device; the plain single-launch solve takes a tenth of it)"""
        lib = self._lib
        lib.sslam_graph_set_option.restype = C.c_int
        lib.sslam_graph_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        rc = lib.sslam_graph_set_option(lib.sslam_slam_graph(self._h), key.encode(), float(value))
        if rc < 0:
            raise SslamError(rc, lib.sslam_last_error().decode())

    def num_edges(self) -> int:
        return int(self._lib.sslam_graph_num_edges(self._lib.sslam_slam_graph(self._h)))

    def graph_vertex(self, vid: int, n: int = 3) -> np.ndarray:
        out = np.zeros(7)
        g = self._lib.sslam_slam_graph(self._h)
        rc = self._lib.sslam_graph_get_vertex(g, vid, out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc < 0:
            raise SslamError(rc, self._lib.sslam_last_error().decode())
        return out[:n]

    def find_matches(self, planes, robot_pose):
        """data_association::find_matches on the device (parity hook; no graph vertices are added)"""
        n = len(planes)
        arr = (Plane * max(n, 1))(*planes)
        out = (Landmark * max(n, 1))()
        rp = np.ascontiguousarray(robot_pose, np.float32).reshape(6)
        self._check(self._lib.sslam_slam_find_matches(self._h, C.cast(arr, C.c_void_p), n, rp.ctypes.data, C.cast(out, C.c_void_p)))
        return [out[k] for k in range(n)]
