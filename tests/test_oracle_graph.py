"""CPU tests of the backend oracle (oracle/oracle_graph.c): it is the checker for the HIP path, so it is
pinned here against (1) finite differences with g2o's numeric scheme, (2) an independent numpy/scipy
restatement, (3) scipy.optimize.least_squares on the same weighted problem, (4) dense inverses,
(5) analytic invariants and (6) the committed golden vectors.  (The reference has no tests: SURVEY.md §4.)"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from semantic_slam_amd.synth import make_graph, pose_compose, pose_inverse
from oracle.oracle import GraphProblem, ET_SE3
from oracle import np_graph

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _full(U):
    return (U + sp.triu(U, 1).T).tocsc()


def _load_g2o(path):
    vt, vf, est, et, ei, ej, meas, info = [], {}, [], [], [], [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "VERTEX_SE3:QUAT":
            vt.append(0); est.append([float(x) for x in t[2:9]])
        elif t[0] == "VERTEX_TRACKXYZ":
            vt.append(1); est.append([float(x) for x in t[2:5]] + [0] * 4)
        elif t[0] == "VERTEX_PLANE":
            vt.append(2); est.append([float(x) for x in t[2:6]] + [0] * 3)
        elif t[0] == "FIX":
            vf[int(t[1])] = 1
        elif t[0].startswith("EDGE"):
            if t[0] == "EDGE_SE3:QUAT":
                d, nz, o, ty = 6, 7, 3, 0
            elif t[0] == "EDGE_SE3_TRACKXYZ":
                d, nz, o, ty = 3, 3, 4, 1
            else:
                d, nz, o, ty = 3, 4, 3, 2
            z = [float(x) for x in t[o:o + nz]]
            up = [float(x) for x in t[o + nz:]]
            W = np.zeros((d, d)); q = 0
            for r in range(d):
                for c in range(r, d):
                    W[r, c] = W[c, r] = up[q]; q += 1
            Wp = np.zeros(36); Wp[:d * d] = W.ravel()
            et.append(ty); ei.append(int(t[1])); ej.append(int(t[2])); meas.append(z + [0] * (7 - nz)); info.append(Wp)
    fixed = [vf.get(v, 0) for v in range(len(vt))]
    return GraphProblem(vt, fixed, est, et, ei, ej, meas, info)


def test_analytic_jacobians_match_g2o_numeric_scheme():
    """central differences, as g2o's BaseBinaryEdge::linearizeOplus does (delta here 1e-7 for FD accuracy)"""
    g = make_graph(30, 6, seed=1)
    gp = GraphProblem.from_synth(g)
    d = 1e-7
    for k in list(range(0, 10)) + list(range(gp.ne - 10, gp.ne)):
        e, Ji, Jj = gp.edge_eval(k)
        is_se3 = gp.etype[k] == ET_SE3
        dim = 6 if is_se3 else 3
        for side, J, vid, vd in ((0, Ji, gp.evi[k], 6), (1, Jj, gp.evj[k], 6 if is_se3 else 3)):
            J = J[:dim * vd].reshape(dim, vd)
            for c in range(vd):
                gpp, gpm = gp.copy(), gp.copy()
                dv = np.zeros(vd); dv[c] = d
                if vd == 6:
                    gpp.est[vid] = np_graph.pose_oplus(gp.est[vid], dv); gpm.est[vid] = np_graph.pose_oplus(gp.est[vid], -dv)
                else:
                    gpp.est[vid, :3] += dv; gpm.est[vid, :3] -= dv
                num = (gpp.edge_eval(k)[0][:dim] - gpm.edge_eval(k)[0][:dim]) / (2 * d)
                assert np.abs(num - J[:, c]).max() < 2e-6


@pytest.mark.parametrize("kind", ["point", "plane"])
def test_linearize_matches_numpy_restatement(kind):
    g = make_graph(60, 12, seed=2, landmark_kind=kind)
    gp = GraphProblem.from_synth(g)
    G = np_graph.NpGraph(g)
    U, b = gp.linearize()
    H, bn = G.build()
    tol = 1e-12 if kind == "point" else 1e-6   # plane Jacobians are finite differences (delta 1e-9) on both sides
    assert abs(_full(U) - H).max() <= tol * abs(H).max()
    assert np.abs(b - bn).max() <= tol * np.abs(bn).max()
    assert gp.chi2() == pytest.approx(G.chi2(), rel=1e-12)
    Hf = _full(U).toarray()
    assert np.allclose(Hf, Hf.T)
    assert np.linalg.eigvalsh(Hf).min() > -1e-8 * np.abs(Hf).max()


def test_hessian_index_follows_g2o_ordering():
    g = make_graph(12, 4, seed=3)
    gp = GraphProblem.from_synth(g, interleave=True)
    h, n = gp.hessian_index()
    off = 0
    for v in range(gp.nv):
        if gp.vfixed[v]:
            assert h[v] == -1
        else:
            assert h[v] == off
            off += 6 if gp.vtype[v] == 0 else 3
    assert n == off


def test_lm_matches_numpy_restatement_and_terminates():
    g = make_graph(120, 24, seed=4)
    gp = GraphProblem.from_synth(g)
    G = np_graph.NpGraph(g)
    st = gp.optimize(12)
    G.optimize(12)
    assert st.iterations == 12
    assert st.chi2_after == pytest.approx(G.chi2(), rel=1e-9)
    assert np.abs(gp.est[:120] - G.poses).max() < 1e-7
    assert np.array_equal(gp.est[0], g.poses_init[0])          # gauge: first vertex fixed (graph_slam.cpp:109-111)
    st2 = gp.optimize(1024)                                      # graph_slam.cpp:205 cap; LM stops on its own
    assert st2.status == 1 and st2.iterations < 1024
    assert st2.chi2_after <= st.chi2_after * (1 + 1e-12)


def test_converged_estimate_matches_scipy_least_squares():
    """Independent solver on the same weighted least-squares problem."""
    from scipy.optimize import least_squares
    g = make_graph(25, 6, seed=5)
    gp = GraphProblem.from_synth(g)
    gp.optimize(60)
    Np, Nl = 25, 6
    Lo = np.linalg.cholesky(g.odom_info[0]).T
    Ll = np.linalg.cholesky(g.lm_info[0]).T
    x0 = np.concatenate([np.zeros(6 * (Np - 1)), np.zeros(3 * Nl)])

    def unpack(x):
        poses = gp.est[:Np].copy()
        poses[1:] = np_graph.pose_oplus(poses[1:], x[:6 * (Np - 1)].reshape(-1, 6))
        lms = gp.est[Np:, :3] + x[6 * (Np - 1):].reshape(-1, 3)
        return poses, lms

    def fun(x):
        poses, lms = unpack(x)
        eo = np_graph.se3_error_jac(poses[g.odom_ij[:, 0]], poses[g.odom_ij[:, 1]], g.odom_z, False) @ Lo.T
        el = np_graph.point_error_jac(poses[g.lm_ij[:, 0]], lms[g.lm_ij[:, 1]], g.lm_z, False) @ Ll.T
        return np.concatenate([eo.ravel(), el.ravel()])

    res = least_squares(fun, x0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    assert np.abs(res.x).max() < 1e-6                 # the oracle's optimum is scipy's optimum
    assert 2 * res.cost == pytest.approx(gp.chi2(), rel=1e-9)


def test_marginals_equal_dense_inverse_blocks():
    g = make_graph(40, 8, seed=6)
    gp = GraphProblem.from_synth(g, interleave=True)
    gp.optimize(8)
    U, _ = gp.linearize()
    Hinv = np.linalg.inv(_full(U).toarray())
    h, _ = gp.hessian_index()
    ids = [int(v) for v in gp.lm_ids]
    blocks = gp.marginals(ids).reshape(-1, 3, 3)
    for v, B in zip(ids, blocks):
        ref = Hinv[h[v]:h[v] + 3, h[v]:h[v] + 3]
        assert np.abs(B - ref).max() <= 1e-9 * np.abs(ref).max()
    # the recursion over the factor (g2o's MarginalCovarianceCholesky, the default) against full triangular solves, poses included
    ids2 = ids + [int(v) for v in gp.pose_ids[1:7]]
    a, b = gp.marginals(ids2), gp.marginals(ids2, by_solves=True)
    assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max()
    o = 9 * len(ids)
    for k, v in enumerate(ids2[len(ids):]):
        ref = Hinv[h[v]:h[v] + 6, h[v]:h[v] + 6]
        assert np.abs(a[o + 36 * k:o + 36 * k + 36].reshape(6, 6) - ref).max() <= 1e-9 * np.abs(ref).max()


def test_noise_free_graph_is_a_fixed_point():
    g = make_graph(40, 8, seed=7, noise_scale=0.0)
    gp = GraphProblem.from_synth(g)
    assert gp.chi2() < 1e-20
    est0 = gp.est.copy()
    gp.optimize(3)
    assert np.abs(gp.est - est0).max() < 1e-9


def test_generator_matches_reference_call_pattern():
    g = make_graph(500, 100, seed=0)
    assert len(g.odom_ij) == 509 and len(g.lm_ij) == 1500
    # odometry information = I/sigma (quirk B3, information_matrix_calculator.cpp:30-32)
    assert np.allclose(np.diag(g.odom_info[0]), [1 / 0.00667] * 3 + [1 / 0.00001] * 3)
    # initial poses are raw integrated odometry (semantic_graph_slam.cpp:120-121)
    assert np.allclose(pose_compose(g.poses_init[10], g.odom_z[10]), g.poses_init[11])
    rel = pose_compose(pose_inverse(g.poses_true[3]), g.poses_true[4])
    assert np.abs(rel[:3] - g.odom_z[3][:3]).max() < 0.2


@pytest.mark.parametrize("kind", ["point", "plane"])
def test_golden_vectors(kind):
    gp = _load_g2o(os.path.join(GOLD, f"graph20_{kind}.g2o"))
    exp = np.load(os.path.join(GOLD, f"graph20_{kind}_expected.npz"))
    assert gp.chi2() == pytest.approx(float(exp["chi2_before"]), rel=1e-13)
    U, b = gp.linearize()
    tol = 1e-13 if kind == "point" else 1e-6
    assert np.abs(U.data - exp["H_upper_data"]).max() <= tol * np.abs(exp["H_upper_data"]).max()
    assert np.array_equal(U.indices, exp["H_upper_indices"]) and np.array_equal(U.indptr, exp["H_upper_indptr"])
    assert np.abs(b - exp["b"]).max() <= tol * np.abs(exp["b"]).max()
    st = gp.optimize(25)
    assert st.chi2_after == pytest.approx(float(exp["chi2_after"]), rel=1e-9)
    assert np.abs(gp.est - exp["estimates"]).max() < 1e-7


@pytest.mark.parametrize("kind", ["point", "plane"])
def test_analytic_invariants(kind):
    """H symmetric positive semi-definite (positive definite with the gauge fixed), chi2 non-increasing over the
    accepted LM steps, the fixed first vertex never moves"""
    gp = GraphProblem.from_synth(make_graph(40, 10, seed=4, landmark_kind=kind), interleave=True)
    U, b = gp.linearize()
    H = _full(U).toarray()
    assert np.abs(H - H.T).max() <= 1e-12 * np.abs(H).max()
    w = np.linalg.eigvalsh(H)
    assert w.min() > 0 and np.isfinite(b).all()
    est0 = gp.est.copy()
    chis = [gp.chi2()]
    for k in range(1, 7):
        g2 = gp.copy()
        st = g2.optimize(k)
        assert st.iterations == k
        chis.append(st.chi2_after)
        assert np.array_equal(g2.est[0], est0[0])          # gauge: vertex 0 is fixed (graph_slam.cpp:109-111)
    assert all(chis[i + 1] <= chis[i] * (1 + 1e-12) for i in range(len(chis) - 1))
    assert chis[-1] < 0.5 * chis[0]


def test_dcs_robust_kernel_follows_g2o_robustify():
    """RobustKernelDCS (the reference means to install it at graph_slam.cpp:155,161 but passes an uninitialised pointer, quirk
    B1): opt-in on the landmark edges.  chi2 = sum rho0(e2) with rho0 = e2 while scale = 2 phi / (phi + e2) >= 1, scale^2 e2 beyond;
    H and b use Omega scaled by rho1.  Checked against a NumPy evaluation from the oracle's own per-edge errors and Jacobians."""
    from oracle import oracle as O
    g = make_graph(40, 8, seed=3)
    gp = GraphProblem.from_synth(g, interleave=True)
    # a gross outlier among the landmark measurements
    k_out = int(np.nonzero(gp.etype == 1)[0][5])
    gp.meas[k_out, :3] += np.array([1.5, -2.0, 1.0])
    phi = 1.0
    chi_plain = gp.chi2()
    U0, b0 = gp.linearize()
    try:
        O.set_dcs(phi)
        chi_dcs = gp.chi2()
        U1, b1 = gp.linearize()
        st = gp.copy().optimize(30)
    finally:
        O.set_dcs(0.0)
    h, n = gp.hessian_index()
    ref_chi = 0.0
    Hd = np.zeros((n, n)); bd = np.zeros(n)
    n_down = 0
    for k in range(gp.ne):
        e, Ji, Jj = gp.edge_eval(k)
        d = 6 if gp.etype[k] == 0 else 3
        dj = 6 if gp.vtype[gp.evj[k]] == 0 else 3
        e, Ji, Jj = e[:d], Ji[:d * 6].reshape(d, 6), Jj[:d * dj].reshape(d, dj)
        W = gp.info[k, :d * d].reshape(d, d)
        e2 = float(e @ W @ e)
        r1 = 1.0
        if gp.etype[k] != 0:
            scale = 2 * phi / (phi + e2)
            if scale < 1:
                r1 = scale * scale; n_down += 1
        ref_chi += r1 * e2
        for (v, J) in ((gp.evi[k], Ji), (gp.evj[k], Jj)):
            if h[v] >= 0:
                bd[h[v]:h[v] + J.shape[1]] -= J.T @ (r1 * W) @ e
        for (va, Ja) in ((gp.evi[k], Ji), (gp.evj[k], Jj)):
            for (vb, Jb) in ((gp.evi[k], Ji), (gp.evj[k], Jj)):
                if h[va] >= 0 and h[vb] >= 0:
                    Hd[h[va]:h[va] + Ja.shape[1], h[vb]:h[vb] + Jb.shape[1]] += Ja.T @ (r1 * W) @ Jb
    assert n_down >= 1 and chi_dcs < chi_plain
    assert abs(chi_dcs - ref_chi) <= 1e-12 * ref_chi
    H1 = (U1 + U1.T).toarray() - np.diag(U1.diagonal())
    assert np.abs(H1 - Hd).max() <= 1e-10 * np.abs(Hd).max() and np.abs(b1 - bd).max() <= 1e-10 * np.abs(bd).max()
    assert np.abs((U1 - U0).toarray()).max() > 0
    assert st.chi2_after <= chi_dcs


def _with_point_point_edges(gp, rng, n_extra=12):
    """append g2o::EdgePointXYZ edges (reference graph_slam.cpp:168-180) between random pairs of point landmarks: measurement = true
    offset + noise, information 4 I (+ a little off-diagonal so that the symmetric-storage paths are exercised)"""
    lms = np.nonzero(gp.vtype == 1)[0]
    pairs = set()
    while len(pairs) < n_extra:
        a, b = rng.choice(lms, 2, replace=False)
        pairs.add((int(a), int(b)))
    pairs = sorted(pairs)
    pairs.append(pairs[0])                                       # a second edge on the same vertex pair
    z = np.zeros((len(pairs), 7)); W = np.zeros((len(pairs), 36))
    for k, (a, b) in enumerate(pairs):
        z[k, :3] = gp.est[b, :3] - gp.est[a, :3] + rng.normal(0, 0.05, 3)
        M = 4.0 * np.eye(3); M[0, 1] = M[1, 0] = 0.3
        W[k, :9] = M.ravel()
    return GraphProblem(gp.vtype, gp.vfixed, gp.est, np.concatenate([gp.etype, np.full(len(pairs), 3, np.int32)]),
                        np.concatenate([gp.evi, [p[0] for p in pairs]]), np.concatenate([gp.evj, [p[1] for p in pairs]]),
                        np.vstack([gp.meas, z]), np.vstack([gp.info, W]))


def test_point_point_edge_follows_g2o_edge_pointxyz():
    """EdgePointXYZ: e = (p2 - p1) - z, Jacobians -I / +I: the oracle's analytic Jacobians equal central differences, the normal
    equations equal a dense J^T W J assembly, and LM drives the chi2 down"""
    rng = np.random.default_rng(9)
    gp = _with_point_point_edges(GraphProblem.from_synth(make_graph(30, 8, seed=5), interleave=True), rng)
    h, n = gp.hessian_index()
    Hd = np.zeros((n, n)); bd = np.zeros(n)
    for k in range(gp.ne):
        e, Ji, Jj = gp.edge_eval(k)
        d = 6 if gp.etype[k] == 0 else 3
        di = 6 if gp.vtype[gp.evi[k]] == 0 else 3
        dj = 6 if gp.vtype[gp.evj[k]] == 0 else 3
        e, Ji, Jj = e[:d], Ji[:d * di].reshape(d, di), Jj[:d * dj].reshape(d, dj)
        if gp.etype[k] == 3:
            assert np.array_equal(Ji, -np.eye(3)) and np.array_equal(Jj, np.eye(3))
            assert np.allclose(e, gp.est[gp.evj[k], :3] - gp.est[gp.evi[k], :3] - gp.meas[k, :3])
        W = gp.info[k, :d * d].reshape(d, d)
        for (va, Ja) in ((gp.evi[k], Ji), (gp.evj[k], Jj)):
            if h[va] >= 0:
                bd[h[va]:h[va] + Ja.shape[1]] -= Ja.T @ W @ e
            for (vb, Jb) in ((gp.evi[k], Ji), (gp.evj[k], Jj)):
                if h[va] >= 0 and h[vb] >= 0:
                    Hd[h[va]:h[va] + Ja.shape[1], h[vb]:h[vb] + Jb.shape[1]] += Ja.T @ W @ Jb
    U, b = gp.linearize()
    H = (U + U.T).toarray() - np.diag(U.diagonal())
    assert np.abs(H - Hd).max() <= 1e-10 * np.abs(Hd).max() and np.abs(b - bd).max() <= 1e-10 * np.abs(bd).max()
    c0 = gp.chi2()
    st = gp.optimize(20)
    assert st.chi2_after < c0
