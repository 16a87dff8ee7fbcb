"""Host-side mirror of ``ps_graph_slam::GraphSLAM`` over the C-ABI (include/sslam.h).

Method names, argument meaning and return conventions follow the reference class
(reference include/ps_graph_slam/graph_slam.hpp:35-152, src/ps_graph_slam/graph_slam.cpp):
``add_se3_node``, ``add_point_xyz_node``, ``add_se3_edge``, ``add_se3_point_xyz_edge``,
``optimize`` (returns ``False`` iff the graph has fewer than 10 edges, graph_slam.cpp:184-186),
``computeLandmarkMarginals``, ``save``.  Vertex handles are plain integer ids.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from ._lib import load_library, OptStats

ERR_TOO_FEW_EDGES = -5


class SslamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"sslam error {code}: {msg}")
        self.code = code


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _check(lib, rc: int) -> int:
    if rc < 0:
        raise SslamError(rc, lib.sslam_last_error().decode())
    return rc


def _pose7(pose) -> np.ndarray:
    """Accept [t(3), q(x,y,z,w)] or a 4x4 / 3x4 isometry (Eigen::Isometry3d in the reference)."""
    a = np.asarray(pose, np.float64)
    if a.shape == (7,):
        return np.ascontiguousarray(a)
    if a.shape in ((4, 4), (3, 4)):
        from .synth import quat_from_matrix
        return np.ascontiguousarray(np.concatenate([a[:3, 3], quat_from_matrix(a[:3, :3])]))
    raise ValueError("pose must be 7 numbers [t, q(xyzw)] or a 4x4 isometry")


class GraphSLAM:
    """``ps_graph_slam::GraphSLAM`` on one MI355X (graph_slam.cpp:40-97)."""

    def __init__(self, verbose: bool = False, device: int = 0):
        self._lib = load_library()
        self.verbose_ = verbose
        self._h = self._lib.sslam_graph_create(device)
        self.last_stats: OptStats | None = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sslam_graph_destroy(h)

    # -- vertices ----------------------------------------------------------------------------
    def add_se3_node(self, pose, fixed: int = -1) -> int:
        """graph_slam.cpp:104-115; the first vertex of the graph is fixed (``fixed=-1``)."""
        return _check(self._lib, self._lib.sslam_graph_add_vertex_se3(self._h, _dptr(_pose7(pose)), fixed))

    def add_point_xyz_node(self, xyz) -> int:
        """graph_slam.cpp:127-134"""
        a = np.ascontiguousarray(xyz, np.float64).reshape(3)
        return _check(self._lib, self._lib.sslam_graph_add_vertex_point(self._h, _dptr(a)))

    def add_plane_node(self, plane_coeffs) -> int:
        """graph_slam.cpp:117-125 (commented out upstream; g2o::VertexPlane)"""
        a = np.ascontiguousarray(plane_coeffs, np.float64).reshape(4)
        return _check(self._lib, self._lib.sslam_graph_add_vertex_plane(self._h, _dptr(a)))

    # -- edges -------------------------------------------------------------------------------
    def add_se3_edge(self, v1: int, v2: int, relative_pose, information_matrix) -> int:
        """graph_slam.cpp:136-148 (information must be 6x6)"""
        w = np.ascontiguousarray(information_matrix, np.float64).reshape(36)
        return _check(self._lib, self._lib.sslam_graph_add_edge_se3(self._h, v1, v2, _dptr(_pose7(relative_pose)), _dptr(w)))

    def add_se3_point_xyz_edge(self, v_se3: int, v_xyz: int, xyz, information_matrix) -> int:
        """graph_slam.cpp:150-166 (information must be 3x3)"""
        z = np.ascontiguousarray(xyz, np.float64).reshape(3)
        w = np.ascontiguousarray(information_matrix, np.float64).reshape(9)
        return _check(self._lib, self._lib.sslam_graph_add_edge_se3_point(self._h, v_se3, v_xyz, _dptr(z), _dptr(w)))

    def add_se3_plane_edge(self, v_se3: int, v_plane: int, plane_coeffs, information_matrix) -> int:
        """graph_slam.hpp:73-75 (commented out upstream) -> include/g2o/edge_se3_plane.hpp"""
        z = np.ascontiguousarray(plane_coeffs, np.float64).reshape(4)
        w = np.ascontiguousarray(information_matrix, np.float64).reshape(9)
        return _check(self._lib, self._lib.sslam_graph_add_edge_se3_plane(self._h, v_se3, v_plane, _dptr(z), _dptr(w)))

    def add_point_xyz_point_xyz_edge(self, v1_xyz: int, v2_xyz: int, xyz, information_matrix) -> int:
        """graph_slam.cpp:168-180: g2o::EdgePointXYZ between two point landmarks, measurement = p2 - p1"""
        z = np.ascontiguousarray(xyz, np.float64).reshape(3)
        w = np.ascontiguousarray(information_matrix, np.float64).reshape(9)
        return _check(self._lib, self._lib.sslam_graph_add_edge_point_point(self._h, v1_xyz, v2_xyz, _dptr(z), _dptr(w)))

    # -- queries -----------------------------------------------------------------------------
    def num_vertices(self) -> int:
        return self._lib.sslam_graph_num_vertices(self._h)

    def num_edges(self) -> int:
        return self._lib.sslam_graph_num_edges(self._h)

    def estimate(self, vid: int) -> np.ndarray:
        out = np.zeros(7)
        n = _check(self._lib, self._lib.sslam_graph_get_vertex(self._h, vid, _dptr(out)))
        return out[:n].copy()

    def set_estimate(self, vid: int, est) -> None:
        a = np.zeros(7)
        e = np.asarray(est, np.float64).ravel()
        a[:len(e)] = e
        _check(self._lib, self._lib.sslam_graph_set_vertex(self._h, vid, _dptr(a)))

    def hessian_index(self, vid: int) -> int:
        return self._lib.sslam_graph_hessian_index(self._h, vid)

    def set_option(self, key: str, value: float) -> None:
        _check(self._lib, self._lib.sslam_graph_set_option(self._h, key.encode(), float(value)))

    def chi2(self) -> float:
        c = C.c_double(0)
        _check(self._lib, self._lib.sslam_graph_chi2(self._h, C.byref(c)))
        return c.value

    # -- the hot entry ------------------------------------------------------------------------
    def optimize(self, max_iterations: int = 1024) -> bool:
        """graph_slam.cpp:182-219. Returns False iff the graph has < 10 edges."""
        st = OptStats()
        rc = self._lib.sslam_graph_optimize(self._h, max_iterations, C.byref(st))
        self.last_stats = st
        if rc == ERR_TOO_FEW_EDGES:
            return False
        _check(self._lib, rc)
        if self.verbose_:
            print(f"iterations: {st.iterations}\nchi2: (before){st.chi2_before} -> (after){st.chi2_after}\n"
                  f"time: {st.seconds:.3f}[sec]")
        return True

    def computeLandmarkMarginals(self, vert_ids: Sequence[int]):
        """graph_slam.cpp:221-234: diagonal blocks of H^-1 for the listed vertices."""
        ids = np.ascontiguousarray(vert_ids, np.int32)
        dims = [7 - 1 if len(self.estimate(int(v))) == 7 else 3 for v in ids]
        out = np.zeros(int(sum(d * d for d in dims)))
        _check(self._lib, self._lib.sslam_graph_marginals(self._h, ids.ctypes.data_as(C.POINTER(C.c_int)), len(ids), _dptr(out)))
        blocks, o = [], 0
        for d in dims:
            blocks.append(out[o:o + d * d].reshape(d, d).copy())
            o += d * d
        return blocks

    def computeMarginals(self, vert_pairs_vec):
        """graph_slam.cpp:221-234 with the reference's own argument: (row, col) pairs of ``hessian_index`` values
        (semantic_graph_slam.cpp:186-191).  Returns {(row, col): block of H^-1}."""
        pairs = np.ascontiguousarray(vert_pairs_vec, np.int32).reshape(-1, 2)
        dim_of = {}
        for v in range(self.num_vertices()):
            h = self.hessian_index(v)
            if h >= 0:
                dim_of[h] = 6 if len(self.estimate(v)) == 7 else 3
        dims = [(dim_of[int(r)], dim_of[int(c)]) for r, c in pairs]
        out = np.zeros(int(sum(a * b for a, b in dims)))
        _check(self._lib, self._lib.sslam_graph_marginals_by_hessian_index(self._h, pairs.ctypes.data_as(C.POINTER(C.c_int)), len(pairs), _dptr(out)))
        res, o = {}, 0
        for (r, c), (a, b) in zip(pairs, dims):
            res[(int(r), int(c))] = out[o:o + a * b].reshape(a, b).copy()
            o += a * b
        return res

    def save(self, filename: str) -> None:
        """graph_slam.cpp:236-239 (g2o text format)"""
        _check(self._lib, self._lib.sslam_graph_save_g2o(self._h, filename.encode()))

    def load(self, filename: str) -> None:
        _check(self._lib, self._lib.sslam_graph_load_g2o(self._h, filename.encode()))

    # -- parity / measurement hooks -------------------------------------------------------------
    def linearize(self):
        """Normal equations at the current estimates: (H upper-triangular scipy CSC, b), g2o order."""
        import scipy.sparse as sp
        dim = C.c_int(0)
        nnz = C.c_int64(0)
        _check(self._lib, self._lib.sslam_graph_linearize(self._h, C.byref(dim), C.byref(nnz), None, None, None, None))
        rows = np.zeros(nnz.value, np.int32); cols = np.zeros(nnz.value, np.int32)
        vals = np.zeros(nnz.value); b = np.zeros(dim.value)
        _check(self._lib, self._lib.sslam_graph_linearize(self._h, C.byref(dim), C.byref(nnz), rows.ctypes.data, cols.ctypes.data,
                                                          vals.ctypes.data, b.ctypes.data))
        U = sp.coo_matrix((vals, (rows, cols)), shape=(dim.value, dim.value)).tocsc()
        return U, b

    def solve(self, lam: float):
        dim = C.c_int(0); nnz = C.c_int64(0)
        _check(self._lib, self._lib.sslam_graph_linearize(self._h, C.byref(dim), C.byref(nnz), None, None, None, None))
        x = np.zeros(dim.value)
        its = C.c_int64(0)
        _check(self._lib, self._lib.sslam_graph_solve(self._h, float(lam), _dptr(x), C.byref(its)))
        return x, its.value

    def oplus(self, dx) -> None:
        a = np.ascontiguousarray(dx, np.float64)
        _check(self._lib, self._lib.sslam_graph_oplus(self._h, _dptr(a)))

    # -- bulk construction helpers ----------------------------------------------------------------
    @classmethod
    def from_problem(cls, gp, device: int = 0) -> "GraphSLAM":
        """Build through the per-vertex/per-edge C-ABI from flat arrays (oracle.GraphProblem layout)."""
        G = cls(False, device)
        for v in range(gp.nv):
            t = int(gp.vtype[v])
            if t == 0:
                vid = G.add_se3_node(gp.est[v], fixed=int(gp.vfixed[v]))
            elif t == 1:
                vid = G.add_point_xyz_node(gp.est[v, :3])
            else:
                vid = G.add_plane_node(gp.est[v, :4])
            assert vid == v
        for k in range(gp.ne):
            t = int(gp.etype[k])
            i, j = int(gp.evi[k]), int(gp.evj[k])
            if t == 0:
                G.add_se3_edge(i, j, gp.meas[k], gp.info[k].reshape(6, 6))
            elif t == 1:
                G.add_se3_point_xyz_edge(i, j, gp.meas[k, :3], gp.info[k, :9].reshape(3, 3))
            elif t == 3:
                G.add_point_xyz_point_xyz_edge(i, j, gp.meas[k, :3], gp.info[k, :9].reshape(3, 3))
            else:
                G.add_se3_plane_edge(i, j, gp.meas[k, :4], gp.info[k, :9].reshape(3, 3))
        return G

    @classmethod
    def from_synth(cls, g, device: int = 0) -> "GraphSLAM":
        """Build a graph from a synth.SynthGraph: poses 0..Np-1 (the first fixed, graph_slam.cpp:109-111), then landmarks."""
        G = cls(False, device)
        Np = g.n_poses
        for i in range(Np):
            G.add_se3_node(g.poses_init[i])
        plane = g.landmark_kind == "plane"
        for l in range(g.n_landmarks):
            (G.add_plane_node if plane else G.add_point_xyz_node)(g.lms_init[l])
        for k in range(len(g.odom_ij)):
            G.add_se3_edge(int(g.odom_ij[k, 0]), int(g.odom_ij[k, 1]), g.odom_z[k], g.odom_info[k])
        for k in range(len(g.lm_ij)):
            (G.add_se3_plane_edge if plane else G.add_se3_point_xyz_edge)(int(g.lm_ij[k, 0]), Np + int(g.lm_ij[k, 1]), g.lm_z[k], g.lm_info[k])
        return G

    def estimates(self) -> np.ndarray:
        n = self.num_vertices()
        out = np.zeros((n, 7))
        for v in range(n):
            e = self.estimate(v)
            out[v, :len(e)] = e
        return out


class GraphBatch:
    """Device-resident batch of independent graphs optimised together (MI355X extension)."""

    def __init__(self, graphs: Sequence[GraphSLAM], streams: int = 0):
        """streams > 1: a stream group (sslam_batch_create_streams) -- the graphs split into that many parts, each on its own HIP stream
        and host thread; 0: sslam_batch_create (one stream unless SSLAM_BATCH_STREAMS says otherwise)"""
        self._lib = load_library()
        self.graphs = list(graphs)
        arr = (C.c_void_p * len(self.graphs))(*[g._h for g in self.graphs])
        if streams > 0:
            self._h = self._lib.sslam_batch_create_streams(arr, len(self.graphs), int(streams))
        else:
            self._h = self._lib.sslam_batch_create(arr, len(self.graphs))
        if not self._h:
            raise SslamError(-1, self._lib.sslam_last_error().decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sslam_batch_destroy(h)

    def upload(self):
        _check(self._lib, self._lib.sslam_batch_upload(self._h))

    def download(self):
        _check(self._lib, self._lib.sslam_batch_download(self._h))

    def optimize(self, max_iterations: int):
        st = (OptStats * len(self.graphs))()
        _check(self._lib, self._lib.sslam_batch_optimize(self._h, max_iterations, st))
        return list(st)

    # -- edge-sharded mode (SURVEY 8e mode E) ----------------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        """RCCL communicator + edge shard of this rank (every rank holds the whole batch); see distributed.init_edge_sharded."""
        _check(self._lib, self._lib.sslam_batch_comm_init(self._h, unique_id, rank, world))

    def set_edge_shard(self, rank: int, world: int) -> None:
        """install the edge shard WITHOUT a communicator: linearize_hb() then returns this rank's partial system (parity hook)"""
        _check(self._lib, self._lib.sslam_batch_set_edge_shard(self._h, rank, world))

    def linearize_hb(self) -> np.ndarray:
        """[H values || b] of the batch at the current estimates (partial if an edge shard is installed)"""
        n = int(self._lib.sslam_batch_linearize_hb(self._h, None, 0))
        _check(self._lib, min(n, 0))
        out = np.zeros(n)
        _check(self._lib, int(self._lib.sslam_batch_linearize_hb(self._h, _dptr(out), int(n))))
        return out

    def time_linearize(self, repeats: int = 20) -> float:
        ms = C.c_double(0)
        _check(self._lib, self._lib.sslam_batch_time_linearize(self._h, repeats, C.byref(ms)))
        return ms.value

    def time_solver(self, repeats: int = 5):
        """(factor ms, backward-solve ms) of one full-batch factorisation + solve, hipEvents on the batch's stream"""
        f = C.c_double(0); s = C.c_double(0)
        _check(self._lib, self._lib.sslam_batch_time_solver(self._h, repeats, C.byref(f), C.byref(s)))
        return f.value, s.value

    def linearize_bytes(self) -> int:
        return int(self._lib.sslam_batch_linearize_bytes(self._h))

    def info(self, key: str) -> float:
        v = C.c_double(0)
        _check(self._lib, self._lib.sslam_batch_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_profiling(self, on: bool):
        _check(self._lib, self._lib.sslam_batch_set_profiling(self._h, 1 if on else 0))

    def kernel_time(self, name: str):
        ms = C.c_double(0); n = C.c_int64(0)
        _check(self._lib, self._lib.sslam_batch_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
