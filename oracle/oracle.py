"""TEST INFRASTRUCTURE — ctypes binding of the plain-C oracle (oracle/oracle_graph.c, oracle_seg.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VT_SE3, VT_POINT, VT_PLANE = 0, 1, 2
ET_SE3, ET_SE3_POINT, ET_SE3_PLANE, ET_POINT_POINT = 0, 1, 2, 3


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_graph.c", "oracle_seg.c") if os.path.exists(os.path.join(_HERE, f))]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


class OgProblem(C.Structure):
    _fields_ = [("nv", C.c_int), ("ne", C.c_int),
                ("vtype", C.c_void_p), ("vfixed", C.c_void_p), ("est", C.c_void_p),
                ("etype", C.c_void_p), ("evi", C.c_void_p), ("evj", C.c_void_p),
                ("meas", C.c_void_p), ("info", C.c_void_p)]


class OgStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("chi2_before", C.c_double),
                ("chi2_after", C.c_double), ("lambda_", C.c_double), ("seconds", C.c_double),
                ("seconds_linearize", C.c_double), ("seconds_solve", C.c_double), ("status", C.c_int)]


def set_dcs(phi: float) -> None:
    """g2o::RobustKernelDCS(delta = phi) on the landmark edges of every problem evaluated from now on (0 = no kernel, the default)"""
    f = lib().og_set_dcs
    f.argtypes = [C.c_double]; f.restype = None
    f(float(phi))


class GraphProblem:
    """Flat arrays describing a graph in the oracle's layout (vertex ids = array index)."""

    def __init__(self, vtype, vfixed, est, etype, evi, evj, meas, info):
        self.vtype = np.ascontiguousarray(vtype, np.int32)
        self.vfixed = np.ascontiguousarray(vfixed, np.int32)
        self.est = np.ascontiguousarray(est, np.float64).reshape(-1, 7).copy()
        self.etype = np.ascontiguousarray(etype, np.int32)
        self.evi = np.ascontiguousarray(evi, np.int32)
        self.evj = np.ascontiguousarray(evj, np.int32)
        self.meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 7)
        self.info = np.ascontiguousarray(info, np.float64).reshape(-1, 36)
        self.nv = len(self.vtype)
        self.ne = len(self.etype)

    @staticmethod
    def from_synth(g, interleave: bool = False) -> "GraphProblem":
        """Vertex ids: poses 0..Np-1 then landmarks (or interleaved in first-seen order, the way
        the reference's orchestrator creates them, semantic_graph_slam.cpp:104-179)."""
        Np, Nl = g.n_poses, g.n_landmarks
        kind = VT_POINT if g.landmark_kind == "point" else VT_PLANE
        ekind = ET_SE3_POINT if g.landmark_kind == "point" else ET_SE3_PLANE
        if not interleave:
            pid = np.arange(Np); lid = Np + np.arange(Nl)
        else:
            first = np.full(Nl, Np, np.int64)
            np.minimum.at(first, g.lm_ij[:, 1], g.lm_ij[:, 0])
            keys = np.concatenate([np.arange(Np) * 2.0, first * 2.0 + 1.0])
            order = np.argsort(keys, kind="stable")
            ids = np.empty(Np + Nl, np.int64); ids[order] = np.arange(Np + Nl)
            pid = ids[:Np]; lid = ids[Np:]
        nv = Np + Nl
        vtype = np.zeros(nv, np.int32); vtype[lid] = kind
        vfixed = np.zeros(nv, np.int32); vfixed[pid[0]] = 1
        est = np.zeros((nv, 7)); est[pid] = g.poses_init; est[lid, :g.lms_init.shape[1]] = g.lms_init
        Eo, El = len(g.odom_ij), len(g.lm_ij)
        etype = np.concatenate([np.full(Eo, ET_SE3, np.int32), np.full(El, ekind, np.int32)])
        evi = np.concatenate([pid[g.odom_ij[:, 0]], pid[g.lm_ij[:, 0]]])
        evj = np.concatenate([pid[g.odom_ij[:, 1]], lid[g.lm_ij[:, 1]]])
        meas = np.zeros((Eo + El, 7)); meas[:Eo] = g.odom_z; meas[Eo:, :g.lm_z.shape[1]] = g.lm_z
        info = np.zeros((Eo + El, 36)); info[:Eo] = g.odom_info.reshape(Eo, 36); info[Eo:, :9] = g.lm_info.reshape(El, 9)
        gp = GraphProblem(vtype, vfixed, est, etype, evi, evj, meas, info)
        gp.pose_ids = np.asarray(pid); gp.lm_ids = np.asarray(lid)
        return gp

    def c_struct(self) -> OgProblem:
        p = OgProblem()
        p.nv, p.ne = self.nv, self.ne
        for name in ("vtype", "vfixed", "est", "etype", "evi", "evj", "meas", "info"):
            setattr(p, name, getattr(self, name).ctypes.data)
        return p

    def copy(self) -> "GraphProblem":
        gp = GraphProblem(self.vtype, self.vfixed, self.est, self.etype, self.evi, self.evj, self.meas, self.info)
        for a in ("pose_ids", "lm_ids"):
            if hasattr(self, a):
                setattr(gp, a, getattr(self, a))
        return gp

    # ---- oracle entry points -------------------------------------------------------------
    def chi2(self) -> float:
        f = lib().og_chi2; f.restype = C.c_double
        p = self.c_struct()
        return f(C.byref(p))

    def hessian_index(self):
        h = np.zeros(self.nv, np.int32)
        p = self.c_struct()
        n = lib().og_hessian_index(C.byref(p), h.ctypes.data_as(C.c_void_p))
        return h, n

    def edge_eval(self, k):
        e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
        p = self.c_struct()
        lib().og_edge_eval(C.byref(p), C.c_int(k), e.ctypes.data_as(C.c_void_p), Ji.ctypes.data_as(C.c_void_p), Jj.ctypes.data_as(C.c_void_p))
        return e, Ji, Jj

    def linearize(self):
        """Upper-triangular CSC (Ap, Ai, Ax) + b in g2o hessian-index order."""
        import scipy.sparse as sp
        p = self.c_struct()
        n = C.c_int(0)
        nnz = lib().og_linearize(C.byref(p), C.byref(n), None, None, None, None)
        Ap = np.zeros(n.value + 1, np.int32); Ai = np.zeros(nnz, np.int32); Ax = np.zeros(nnz); b = np.zeros(n.value)
        lib().og_linearize(C.byref(p), C.byref(n), Ap.ctypes.data_as(C.c_void_p), Ai.ctypes.data_as(C.c_void_p),
                           Ax.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        U = sp.csc_matrix((Ax, Ai, Ap), shape=(n.value, n.value))
        return U, b

    def solve(self, lam: float):
        _, n = self.hessian_index()
        x = np.zeros(n)
        p = self.c_struct()
        rc = lib().og_solve(C.byref(p), C.c_double(lam), x.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("oracle Cholesky failed")
        return x

    def oplus(self, dx):
        h, n = self.hessian_index()
        dx = np.ascontiguousarray(dx, np.float64)
        p = self.c_struct()
        lib().og_oplus(C.byref(p), h.ctypes.data_as(C.c_void_p), dx.ctypes.data_as(C.c_void_p), self.est.ctypes.data_as(C.c_void_p))

    def optimize(self, max_iters: int = 1024) -> OgStats:
        st = OgStats()
        p = self.c_struct()
        lib().og_optimize(C.byref(p), C.c_int(max_iters), C.byref(st))
        return st

    def marginals(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        dims = np.where(self.vtype[ids] == VT_SE3, 6, 3)
        out = np.zeros(int((dims * dims).sum()))
        p = self.c_struct()
        rc = lib().og_marginals(C.byref(p), ids.ctypes.data_as(C.c_void_p), C.c_int(len(ids)), out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("oracle marginals: H not positive definite")
        return out


# ---- frontend oracle entry (oracle_seg.c) ------------------------------------------------------
def segment_frame(frame, params, want_products: bool = False):
    """CPU oracle of point_cloud_segmentation::segmentallPointCloudData on one synthetic frame.
    `params` / the returned plane records use the C-ABI struct layouts (sslam_seg_params / sslam_plane),
    which oracle_seg.c shares.  Returns (planes, normals[npix,4] | None, labels[npix] | None)."""
    from semantic_slam_amd.segmentation import Plane   # ctypes mirror of sslam_plane (layout only)
    L = lib()
    out = (Plane * 512)()
    npix = int(sum(int(b["width"]) * int(b["height"]) for b in frame.boxes))
    nrm = np.zeros((npix, 4), np.float32) if want_products else None
    lab = np.zeros(npix, np.int32) if want_products else None
    L.os_segment.restype = C.c_int
    n = L.os_segment(C.byref(params), frame.cloud.ctypes.data_as(C.c_void_p), frame.width, frame.height, frame.point_step,
                     frame.row_step, frame.offsets[0], frame.offsets[1], frame.offsets[2],
                     frame.boxes.ctypes.data_as(C.c_void_p), len(frame.boxes), frame.robot_pose.ctypes.data_as(C.c_void_p),
                     C.c_float(frame.cam_angle), out, 512,
                     nrm.ctypes.data_as(C.c_void_p) if want_products else None,
                     lab.ctypes.data_as(C.c_void_p) if want_products else None)
    return [out[k] for k in range(n)], nrm, lab
