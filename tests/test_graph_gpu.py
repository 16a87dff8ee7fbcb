"""GPU parity of the ps_graph_slam backend: HIP path (through the C-ABI) vs the CPU oracle.

Reference semantics under test: GraphSLAM::optimize (reference src/ps_graph_slam/graph_slam.cpp:182-219)
and the g2o types it instantiates.  Tolerances: analytic edges H/b 1e-11 relative; plane edges
(numeric Jacobian, delta=1e-9) 1e-5; converged estimates <= 1e-4 relative (north_star), chi2 1e-6.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from semantic_slam_amd.synth import make_graph
from oracle.oracle import GraphProblem

pytestmark = pytest.mark.gpu


def _full(U):
    return (U + sp.triu(U, 1).T).tocsc()


@pytest.mark.parametrize("kind,interleave,tol", [("point", False, 1e-11), ("point", True, 1e-11), ("plane", False, 2e-5)])
def test_linearize_matches_oracle(gpu_lib, kind, interleave, tol):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(60, 12, seed=3, landmark_kind=kind)
    gp = GraphProblem.from_synth(g, interleave=interleave)
    G = GraphSLAM.from_problem(gp)
    U, b = G.linearize()
    Uo, bo = gp.linearize()
    assert U.shape == Uo.shape
    scale = abs(Uo).max()
    assert abs(_full(U) - _full(Uo)).max() <= tol * scale
    assert np.abs(b - bo).max() <= tol * max(1.0, np.abs(bo).max())
    assert abs(G.chi2() - gp.chi2()) <= 1e-12 * gp.chi2()
    h, n = gp.hessian_index()
    assert [G.hessian_index(v) for v in range(gp.nv)] == list(h)


@pytest.mark.parametrize("deterministic", [1, 2, 0])
def test_linearize_with_repeated_edges(gpu_lib, deterministic):
    """Two edges on the same vertex pair (a repeated loop closure / a landmark matched twice in one
    keyframe) share one off-diagonal block; gather-form and atomic Jacobian builds must both sum them."""
    from semantic_slam_amd import GraphSLAM
    g = make_graph(60, 12, seed=8)
    gp0 = GraphProblem.from_synth(g, interleave=True)
    Eo = len(g.odom_ij)
    dup = [3, 17, Eo + 5, Eo + 40, 3]
    rng = np.random.default_rng(1)
    meas = np.concatenate([gp0.meas, gp0.meas[dup] + rng.normal(0, 1e-3, (len(dup), 7))])
    meas[:, 3:7] /= np.where(np.linalg.norm(meas[:, 3:7], axis=1, keepdims=True) > 0.5, np.linalg.norm(meas[:, 3:7], axis=1, keepdims=True), 1.0)
    gp = GraphProblem(gp0.vtype, gp0.vfixed, gp0.est, np.concatenate([gp0.etype, gp0.etype[dup]]),
                      np.concatenate([gp0.evi, gp0.evi[dup]]), np.concatenate([gp0.evj, gp0.evj[dup]]),
                      meas, np.concatenate([gp0.info, gp0.info[dup]]))
    G = GraphSLAM.from_problem(gp)
    G.set_option("deterministic", deterministic)
    U, b = G.linearize()
    Uo, bo = gp.linearize()
    assert abs(_full(U) - _full(Uo)).max() <= 1e-11 * abs(Uo).max()
    assert np.abs(b - bo).max() <= 1e-11 * np.abs(bo).max()
    assert G.optimize(5)
    st = gp.optimize(5)
    assert G.last_stats.chi2_after == pytest.approx(st.chi2_after, rel=1e-6)


def test_jacobian_build_is_bitwise_deterministic(gpu_lib):
    from semantic_slam_amd import GraphSLAM
    gp = GraphProblem.from_synth(make_graph(200, 40, seed=9))
    G = GraphSLAM.from_problem(gp)
    U1, b1 = G.linearize()
    U2, b2 = G.linearize()
    assert np.array_equal(U1.data, U2.data) and np.array_equal(b1, b2)


def test_oplus_matches_oracle(gpu_lib):
    from semantic_slam_amd import GraphSLAM
    for kind in ("point", "plane"):
        g = make_graph(40, 9, seed=5, landmark_kind=kind)
        gp = GraphProblem.from_synth(g, interleave=True)
        G = GraphSLAM.from_problem(gp)
        _, n = gp.hessian_index()
        dx = np.random.default_rng(0).normal(0, 0.05, n)
        G.oplus(dx)
        gp.oplus(dx)
        assert np.abs(G.estimates() - gp.est).max() < 1e-13


@pytest.mark.parametrize("lam", [5.0, 1e-3])
def test_pcg_solve_matches_oracle_cholesky(gpu_lib, lam):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(120, 25, seed=1)
    gp = GraphProblem.from_synth(g)
    G = GraphSLAM.from_problem(gp)
    G.set_option("solver", 0)
    x, its = G.solve(lam)
    xo = gp.solve(lam)
    assert its > 0
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()


@pytest.mark.parametrize("lam", [5.0, 1e-3, 0.0])
def test_cholesky_solve_matches_oracle_cholesky(gpu_lib, lam):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(150, 30, seed=7)
    gp = GraphProblem.from_synth(g, interleave=True)
    G = GraphSLAM.from_problem(gp)
    G.set_option("solver", 1)
    x, its = G.solve(lam)
    xo = gp.solve(lam)
    assert its == 0
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    # and the residual of the normal equations themselves
    U, b = G.linearize()
    H = _full(U) + lam * sp.identity(U.shape[0])
    assert np.abs(H @ x - b).max() <= 1e-9 * np.abs(b).max()


@pytest.mark.parametrize("kind,solver", [("point", 1), ("plane", 1), ("point", 0), ("plane", 0)])
def test_optimize_small_graph_matches_oracle(gpu_lib, kind, solver):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(100, 20, seed=2, landmark_kind=kind)
    gp = GraphProblem.from_synth(g, interleave=True)
    G = GraphSLAM.from_problem(gp)
    G.set_option("solver", solver)
    assert G.optimize(12) is True
    st = gp.optimize(12)
    s = G.last_stats
    assert s.chi2_before == pytest.approx(st.chi2_before, rel=1e-12)
    assert s.chi2_after == pytest.approx(st.chi2_after, rel=1e-6)
    E, Eo = G.estimates(), gp.est
    assert np.abs(E - Eo).max() <= 1e-4 * np.abs(Eo).max()


@pytest.mark.parametrize("solver", [1, 0])
def test_optimize_S_config_10_iterations(gpu_lib, solver):
    """BASELINE.json configs[1]: 500 poses / 100 landmarks, exactly 10 LM iterations."""
    from semantic_slam_amd import GraphSLAM
    g = make_graph(500, 100, seed=0)
    gp = GraphProblem.from_synth(g)
    G = GraphSLAM.from_problem(gp)
    G.set_option("solver", solver)
    assert G.optimize(10)
    st = gp.optimize(10)
    s = G.last_stats
    assert s.iterations == st.iterations == 10
    assert s.chi2_after == pytest.approx(st.chi2_after, rel=1e-6)
    E, Eo = G.estimates(), gp.est
    assert np.abs(E - Eo).max() <= 1e-4 * np.abs(Eo).max()
    # gauge: the fixed first vertex does not move (graph_slam.cpp:109-111)
    assert np.array_equal(E[0], g.poses_init[0])


def test_optimize_L_config(gpu_lib):
    """BASELINE.json configs[2]: 5000 poses / 1000 landmarks + loop closures; 10 iterations, then to termination."""
    from semantic_slam_amd import GraphSLAM
    g = make_graph(5000, 1000, seed=0)
    gp = GraphProblem.from_synth(g)
    G = GraphSLAM.from_problem(gp)
    assert G.optimize(10)
    st = gp.optimize(10)
    s = G.last_stats
    assert s.iterations == st.iterations == 10
    assert s.chi2_after == pytest.approx(st.chi2_after, rel=1e-6)
    assert np.abs(G.estimates() - gp.est).max() <= 1e-4 * np.abs(gp.est).max()
    # run both to LM termination (cap 1024 as graph_slam.cpp:205): converged estimates must agree
    assert G.optimize(1024)
    st2 = gp.optimize(1024)
    assert G.last_stats.status == 1 and st2.status == 1
    assert G.last_stats.chi2_after == pytest.approx(st2.chi2_after, rel=1e-8)
    assert np.abs(G.estimates() - gp.est).max() <= 1e-4 * np.abs(gp.est).max()


def test_too_few_edges_returns_false(gpu_lib):
    from semantic_slam_amd import GraphSLAM
    G = GraphSLAM()
    a = G.add_se3_node([0, 0, 0, 0, 0, 0, 1])
    b = G.add_se3_node([1, 0, 0, 0, 0, 0, 1])
    G.add_se3_edge(a, b, [1, 0, 0, 0, 0, 0, 1], np.eye(6))
    assert G.optimize() is False          # graph_slam.cpp:184-186
    assert np.array_equal(G.estimate(b), [1, 0, 0, 0, 0, 0, 1])


def test_noise_free_graph_is_a_fixed_point(gpu_lib):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(50, 10, seed=4, noise_scale=0.0)
    gp = GraphProblem.from_synth(g)
    G = GraphSLAM.from_problem(gp)
    assert G.chi2() < 1e-18
    G.optimize(3)
    assert G.last_stats.chi2_after < 1e-18
    E = G.estimates()[:50]
    sign = np.where(np.sum(E[:, 3:] * g.poses_true[:, 3:], axis=1, keepdims=True) < 0, -1.0, 1.0)  # q ~ -q
    assert np.abs(E[:, :3] - g.poses_true[:, :3]).max() < 1e-9
    assert np.abs(sign * E[:, 3:] - g.poses_true[:, 3:]).max() < 1e-9


def test_batch_matches_individual(gpu_lib):
    from semantic_slam_amd import GraphSLAM, GraphBatch
    sizes = [(60, 12), (45, 9), (80, 15)]
    gps = [GraphProblem.from_synth(make_graph(a, b, seed=10 + i)) for i, (a, b) in enumerate(sizes)]
    singles = [GraphSLAM.from_problem(gp) for gp in gps]
    for G in singles:
        G.optimize(8)
    batch_graphs = [GraphSLAM.from_problem(gp) for gp in gps]
    B = GraphBatch(batch_graphs)
    B.upload()
    stats = B.optimize(8)
    B.download()
    for G1, G2, st in zip(singles, batch_graphs, stats):
        assert st.iterations == G1.last_stats.iterations
        assert st.chi2_after == pytest.approx(G1.last_stats.chi2_after, rel=1e-9)
        assert np.abs(G1.estimates() - G2.estimates()).max() < 1e-9


def test_marginals_match_oracle(gpu_lib):
    from semantic_slam_amd import GraphSLAM
    g = make_graph(40, 8, seed=6)
    gp = GraphProblem.from_synth(g, interleave=True)
    G = GraphSLAM.from_problem(gp)
    G.optimize(6)
    gp.est[:] = G.estimates()
    ids = [int(v) for v in gp.lm_ids]
    blocks = G.computeLandmarkMarginals(ids)
    ref = gp.marginals(ids).reshape(-1, 3, 3)
    for a, r in zip(blocks, ref):
        assert np.abs(a - r).max() <= 1e-6 * np.abs(r).max()
