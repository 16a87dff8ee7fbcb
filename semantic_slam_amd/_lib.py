"""ctypes loader for libsslam_hip.so (the C-ABI declared in include/sslam.h).

Fails loudly: if the shared library is missing or cannot be loaded an ImportError/OSError is
raised — there is no eager/CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path() -> str:
    return os.path.join(_PKG, "libsslam_hip.so")


def build_library(force: bool = False) -> str:
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_PKG, "csrc"), "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return library_path()


class OptStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("status", C.c_int), ("host_plan_us", C.c_int),
                ("chi2_before", C.c_double), ("chi2_after", C.c_double), ("lambda_", C.c_double),
                ("seconds", C.c_double), ("solver_iterations", C.c_int64)]

    def __repr__(self):
        return (f"OptStats(iterations={self.iterations}, trials={self.trials}, status={self.status}, "
                f"chi2 {self.chi2_before:.6g} -> {self.chi2_after:.6g}, lambda={self.lambda_:.3g}, "
                f"seconds={self.seconds:.4f}, solver_iterations={self.solver_iterations})")


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C semantic_slam_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, ci, cd, i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
    dp = C.POINTER(C.c_double)
    sig = {
        "sslam_last_error": (C.c_char_p, []),
        "sslam_device_count": (ci, []),
        "sslam_pinned_alloc": (vp, [C.c_size_t]),
        "sslam_pinned_free": (None, [vp]),
        "sslam_graph_create": (vp, [ci]),
        "sslam_graph_destroy": (None, [vp]),
        "sslam_graph_add_vertex_se3": (ci, [vp, dp, ci]),
        "sslam_graph_add_vertex_point": (ci, [vp, dp]),
        "sslam_graph_add_vertex_plane": (ci, [vp, dp]),
        "sslam_graph_add_edge_se3": (ci, [vp, ci, ci, dp, dp]),
        "sslam_graph_add_edge_se3_point": (ci, [vp, ci, ci, dp, dp]),
        "sslam_graph_add_edge_se3_plane": (ci, [vp, ci, ci, dp, dp]),
        "sslam_graph_add_edge_point_point": (ci, [vp, ci, ci, dp, dp]),
        "sslam_graph_num_vertices": (ci, [vp]),
        "sslam_graph_num_edges": (ci, [vp]),
        "sslam_graph_get_vertex": (ci, [vp, ci, dp]),
        "sslam_graph_set_vertex": (ci, [vp, ci, dp]),
        "sslam_graph_hessian_index": (ci, [vp, ci]),
        "sslam_graph_set_option": (ci, [vp, C.c_char_p, cd]),
        "sslam_graph_optimize": (ci, [vp, ci, C.POINTER(OptStats)]),
        "sslam_graph_chi2": (ci, [vp, dp]),
        "sslam_graph_marginals": (ci, [vp, C.POINTER(ci), ci, dp]),
        "sslam_graph_marginals_by_hessian_index": (ci, [vp, C.POINTER(ci), ci, dp]),
        "sslam_graph_save_g2o": (ci, [vp, C.c_char_p]),
        "sslam_graph_load_g2o": (ci, [vp, C.c_char_p]),
        "sslam_graph_linearize": (ci, [vp, C.POINTER(ci), C.POINTER(i64), vp, vp, vp, vp]),
        "sslam_graph_solve": (ci, [vp, cd, dp, C.POINTER(i64)]),
        "sslam_graph_oplus": (ci, [vp, dp]),
        "sslam_debug_plan_create": (vp, [C.POINTER(vp), ci]),
        "sslam_debug_plan_destroy": (None, [vp]),
        "sslam_debug_plan_array": (i64, [vp, C.c_char_p, vp, i64]),
        "sslam_batch_create": (vp, [C.POINTER(vp), ci]),
        "sslam_batch_create_streams": (vp, [C.POINTER(vp), ci, ci]),
        "sslam_batch_destroy": (None, [vp]),
        "sslam_batch_upload": (ci, [vp]),
        "sslam_batch_download": (ci, [vp]),
        "sslam_batch_optimize": (ci, [vp, ci, C.POINTER(OptStats)]),
        "sslam_comm_unique_id": (ci, [C.c_char_p]),
        "sslam_batch_comm_init": (ci, [vp, C.c_char_p, ci, ci]),
        "sslam_batch_set_edge_shard": (ci, [vp, ci, ci]),
        "sslam_batch_linearize_hb": (i64, [vp, dp, i64]),
        "sslam_batch_time_linearize": (ci, [vp, ci, dp]),
        "sslam_batch_time_solver": (ci, [vp, ci, dp, dp]),
        "sslam_batch_linearize_bytes": (i64, [vp]),
        "sslam_batch_info": (ci, [vp, C.c_char_p, dp]),
        "sslam_batch_set_profiling": (ci, [vp, ci]),
        "sslam_batch_kernel_time": (ci, [vp, C.c_char_p, dp, C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
