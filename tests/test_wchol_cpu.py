"""CPU-side pin of the window multifrontal Cholesky (semantic_slam_amd/csrc/wchol_plan.hpp + sslam_wchol.hip): the library runs its
own plan and the SAME per-thread phase functions the HIP kernels are made of through a host executor (sslam_debug_wchol_solve, no GPU
needed: a loop over thread ids per phase instead of a barrier), and the solution of (H + lambda I) x = b is compared with dense
linear algebra on the oracle's normal equations.  Reference anchor: the linear solver behind GraphSLAM::optimize
(reference src/ps_graph_slam/graph_slam.cpp:27,67-73)."""
import ctypes as C
import os

import numpy as np
import pytest

from semantic_slam_amd.synth import make_graph
from oracle.oracle import GraphProblem
from chol_plan_exec import Plan
from test_chol_plan_cpu import _internal_order


def _bind(lib):
    lib.sslam_debug_wchol_solve.restype = C.c_int64
    lib.sslam_debug_wchol_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def _systems(hip_lib, specs):
    """graphs + their normal equations in the device layout of a batch: ([GraphSLAM], [H || b] buffer, per-graph dense (H, b, rows))"""
    from semantic_slam_amd import GraphSLAM
    gps = [GraphProblem.from_synth(make_graph(a, b, seed=sd, **kw), interleave=il) for (a, b, sd, il, kw) in specs]
    Gs = [GraphSLAM.from_problem(gp) for gp in gps]
    plan = Plan(hip_lib, Gs)              # the piece plan's exported structure arrays give the H layout of the batch (pack_H)
    dense, Hs = [], []
    for gp in gps:
        U, b = gp.linearize()
        Hg = (U + U.T).toarray() - np.diag(U.diagonal())
        idx, n = _internal_order(gp)
        dense.append((Hg[np.ix_(idx, idx)], b[idx]))
    # internal row order of the batch: pose rows of all graphs, then landmark rows of all graphs
    npr = [int((gp.vtype[gp.hessian_index()[0] >= 0] == 0).sum()) for gp in gps]
    nlr = [int((gp.vtype[gp.hessian_index()[0] >= 0] != 0).sum()) for gp in gps]
    nP, nL = sum(npr), sum(nlr)
    dim = 6 * nP + 3 * nL
    assert dim == plan.dim
    Hd = np.zeros((dim, dim)); bd = np.zeros(dim)
    rows = []
    p0 = l0 = 0
    for (Hg, bg), a, c in zip(dense, npr, nlr):
        r = np.concatenate([np.arange(6 * p0, 6 * (p0 + a)), 6 * nP + np.arange(3 * l0, 3 * (l0 + c))])
        Hd[np.ix_(r, r)] = Hg; bd[r] = bg
        rows.append(r)
        p0 += a; l0 += c
    Hdev = plan.pack_H(Hd)
    h_even = (plan.h_total + 1) & ~1
    hb = np.zeros(h_even + dim)
    hb[:plan.h_total] = Hdev; hb[h_even:] = bd
    return Gs, hb, dense, rows, dim


def _solve(hip_lib, Gs, hb, lam, dim):
    _bind(hip_lib)
    arr = (C.c_void_p * len(Gs))(*[g._h for g in Gs])
    x = np.zeros(dim); fail = np.zeros(len(Gs), np.int32); stats = np.zeros(8, np.int64)
    lam = np.ascontiguousarray(lam, np.float64)
    need = hip_lib.sslam_debug_wchol_solve(arr, len(Gs), None, 0, None, None, None, None)
    assert need == len(hb)
    rc = hip_lib.sslam_debug_wchol_solve(arr, len(Gs), hb.ctypes.data, len(hb), lam.ctypes.data, x.ctypes.data, fail.ctypes.data, stats.ctypes.data)
    assert rc == need, hip_lib.sslam_last_error()
    return x, fail, stats


@pytest.fixture(scope="module")
def hip_lib():
    from semantic_slam_amd import load_library
    return load_library()


def _check(hip_lib, specs, lams, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = str(v)
    try:
        Gs, hb, dense, rows, dim = _systems(hip_lib, specs)
        x, fail, stats = _solve(hip_lib, Gs, hb, lams, dim)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k)
            else: os.environ[k] = v
    assert not fail.any()
    for (Hg, bg), r, lam in zip(dense, rows, lams):
        xref = np.linalg.solve(Hg + lam * np.eye(len(bg)), bg)
        assert np.abs(x[r] - xref).max() <= 1e-8 * np.abs(xref).max()
    return stats


@pytest.mark.parametrize("form", ["mfma", "valu"])
def test_window_cholesky_solves_small_graphs(hip_lib, form, monkeypatch):
    """both forms of the window update: the matrix-core form (16 x 16 accumulator tiles, panel P P^T) and the VALU form (3 x 3 quarters)"""
    monkeypatch.setenv("SSLAM_WCHOL_VALU", "1" if form == "valu" else "0")
    specs = [(40, 8, 1, False, {}), (60, 12, 2, True, {}), (30, 6, 3, True, dict(landmark_kind="plane")), (25, 5, 4, False, dict(loop_every=5))]
    st = _check(hip_lib, specs, [0.0, 1e-3, 2.5, 10.0])
    assert st[2] >= 4 and st[6] == sum(a - 1 + b for a, b, *_ in specs)     # segments, columns


@pytest.mark.parametrize("form", ["mfma", "valu"])
def test_window_cholesky_S_config_with_both_classes(hip_lib, form, monkeypatch):
    monkeypatch.setenv("SSLAM_WCHOL_VALU", "1" if form == "valu" else "0")
    _S_config(hip_lib)


def _S_config(hip_lib):
    """the S graph has columns with more than 9 off-diagonal blocks: its top runs in the four-wave class, the rest in one-wave segments,
    with update matrices crossing the class border"""
    st = _check(hip_lib, [(500, 100, 0, False, {})], [1e-5 * 3e5])
    assert st[7] >= 1 and st[2] > st[7] >= 1 and st[3] >= 3


@pytest.mark.parametrize("env", [dict(SSLAM_WCHOL_COLCAP0=1, SSLAM_WCHOL_COLCAP1=1), dict(SSLAM_WCHOL_COLCAP0=3, SSLAM_WCHOL_COLCAP1=5),
                                 dict(SSLAM_WCHOL_WSMALL=4), dict(SSLAM_WCHOL_COLCAP0=1000, SSLAM_WCHOL_COLCAP1=4000)])
def test_window_cholesky_under_forced_segment_shapes(hip_lib, env):
    """one column per segment (every update matrix goes through HBM), tiny segments, a tiny one-wave window (most columns pushed into
    the four-wave class), and segments as long as the window allows"""
    _check(hip_lib, [(120, 24, 7, True, {}), (90, 30, 8, False, {})], [0.5, 0.0], env)


def test_window_cholesky_flags_an_indefinite_system(hip_lib):
    Gs, hb, dense, rows, dim = _systems(hip_lib, [(30, 6, 11, False, {}), (30, 6, 12, False, {})])
    x, fail, _ = _solve(hip_lib, Gs, hb, [-1e9, 0.0], dim)           # graph 0: H - 1e9 I is negative definite
    assert fail[0] == 1 and fail[1] == 0
    Hg, bg = dense[1]
    assert np.abs(x[rows[1]] - np.linalg.solve(Hg, bg)).max() <= 1e-8 * np.abs(x[rows[1]]).max()
