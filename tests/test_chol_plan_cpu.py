"""CPU-side pin of the symbolic sparse-Cholesky plan (pieces, work items, update records) that the HIP kernels of
semantic_slam_amd/csrc/sslam_chol.hip execute: the plan is exported through the C-ABI (no GPU needed), walked by a numpy
executor in the kernels' own order (tests/chol_plan_exec.py) and compared with dense linear algebra on the oracle's normal
equations.  Reference anchor: the linear solver behind GraphSLAM::optimize (reference src/ps_graph_slam/graph_slam.cpp:27,67-73)."""
import os

import numpy as np
import pytest

from semantic_slam_amd.synth import make_graph
from oracle.oracle import GraphProblem
from chol_plan_exec import Plan, upd_fields


def _internal_order(gp):
    """g2o hessian offsets -> the backend's internal row order (pose rows, then landmark rows, each by vertex id)."""
    h, n = gp.hessian_index()
    idx = []
    for v in range(gp.nv):
        if gp.vtype[v] == 0 and h[v] >= 0:
            idx += list(range(h[v], h[v] + 6))
    for v in range(gp.nv):
        if gp.vtype[v] != 0 and h[v] >= 0:
            idx += list(range(h[v], h[v] + 3))
    return np.array(idx), n


def _plan_and_system(hip_lib, g, interleave, env=None):
    from semantic_slam_amd import GraphSLAM
    old = {}
    env = dict({"front": 1}, **(env or {}))   # the front tables (front_plan.hpp) ride along with every plan of these tests: _check walks them too
    env = {"SSLAM_CHOL_OPTS": ",".join(f"{k}={v}" for k, v in env.items())} if env else {}   # plan options by field name (chol_plan.hpp CholOpts)
    for k, v in env.items():
        old[k] = os.environ.get(k); os.environ[k] = str(v)
    try:
        gp = GraphProblem.from_synth(g, interleave=interleave)
        G = GraphSLAM.from_problem(gp)
        plan = Plan(hip_lib, [G])
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k)
            else: os.environ[k] = v
    U, b = gp.linearize()
    Hg = (U + U.T).toarray() - np.diag(U.diagonal())
    idx, n = _internal_order(gp)
    assert n == plan.dim and len(idx) == n
    return plan, Hg[np.ix_(idx, idx)], b[idx]


def _check(plan, H, b, lam):
    Hdev = plan.pack_H(H)
    Lval, y, ok = plan.factor(Hdev, b, lam)
    assert ok
    Ld = plan.dense_L(Lval)
    px, py = plan.perm_x_to_y()
    A = H + lam * np.eye(plan.dim)
    Ay = np.zeros_like(A); Ay[np.ix_(py, py)] = A[np.ix_(px, px)]
    scale = np.abs(A).max()
    assert np.abs(Ld @ Ld.T - Ay).max() <= 1e-9 * scale
    assert np.abs(np.triu(Ld, 1)).max() == 0.0
    by = np.zeros(plan.dim); by[py] = b[px]
    yref = np.linalg.solve(Ld, by)
    assert np.abs(y[:plan.dim] - yref).max() <= 1e-9 * max(np.abs(yref).max(), 1e-30)
    x = plan.backward(Lval, y)
    xref = np.linalg.solve(A, b)
    assert np.abs(x - xref).max() <= 1e-7 * np.abs(xref).max()
    y2 = plan.forward_rows(Lval, b)          # the row lists of the multi right-hand-side forward substitution
    assert np.abs(y2 - yref).max() <= 1e-9 * max(np.abs(yref).max(), 1e-30)
    if np.any(plan.piece["pad5"] >= 1):      # the tail and mid pieces' right-looking update lists give the same factor (other summation order)
        assert len(plan.rupd) == int(plan.piece["nu_i"][plan.piece["pad5"] >= 1].sum())
        Lr, yr, okr = plan.factor(Hdev, b, lam, right=True)
        assert okr
        assert np.abs(Lr - Lval).max() <= 1e-10 * max(np.abs(Lval).max(), 1e-30)
        assert np.abs(yr[:plan.dim] - y[:plan.dim]).max() <= 1e-9 * max(np.abs(yref).max(), 1e-30)
    if plan.front:   # the same pieces through the front tables (one blob of relative indices per workgroup): the walk of k_front_pieces / k_front_tail
        assert plan.funz == plan.unz                     # dense lower triangles over the boundary rows: what the record plan allocates block by block
        for right in (True, False):                      # right-looking mid / tail pieces (what the kernels run) and target tiles everywhere
            Lf, yf, okf = plan.factor_front(Hdev, b, lam, right=right)
            assert okf
            assert np.abs(Lf - Lval).max() <= 1e-10 * max(np.abs(Lval).max(), 1e-30)
            assert np.abs(yf[:plan.dim] - y[:plan.dim]).max() <= 1e-9 * max(np.abs(yref).max(), 1e-30)
    return Lval


def _structure_invariants(plan):
    P = plan.piece
    # pieces tile the columns, blocks, L storage and y contiguously
    assert P["c0"][0] == 0 and np.all(P["c0"][1:] == P["c0"][:-1] + P["nc"][:-1]) and P["c0"][-1] + P["nc"][-1] == plan.ncol
    assert np.all(P["lbase"][1:] == P["lbase"][:-1] + P["lsize"][:-1]) and P["lbase"][-1] + P["lsize"][-1] == plan.lnz
    assert np.all(P["y0"][1:] == P["y0"][:-1] + P["ysize"][:-1]) and P["y0"][-1] + P["ysize"][-1] == plan.dim
    # every piece is launched exactly once
    order = plan.piece_order()
    assert sorted(order) == list(range(plan.npiece))
    # internal updates stay inside their piece; the update-matrix blocks a piece absorbs or passes on were written by pieces that
    # run earlier, and never by a piece of the same launch
    rank = np.zeros(plan.npiece, int); rank[order] = np.arange(plan.npiece)
    depth_of = {}
    for l in range(len(plan.plv_ptr) - 1):
        for p in plan.plv_pieces[plan.plv_ptr[l]:plan.plv_ptr[l + 1]]:
            depth_of[int(p)] = l
    writer = {}
    for p in range(plan.npiece):
        pm = P[p]
        for im in plan.uitem[pm["uit0"]:pm["uit0"] + pm["nuit"]]:
            writer[int(im["uoff"])] = p
            if im["uyoff"] >= 0:
                writer[int(im["uyoff"])] = p
    for p in range(plan.npiece):
        pm = P[p]
        for im in plan.item[pm["iit0"]:pm["iit0"] + pm["nit_i"]]:
            assert 0 <= im["u0"] and im["u0"] + im["n"] <= pm["nu_i"]
            for r in plan.upd[pm["iu0"] + im["u0"]:pm["iu0"] + im["u0"] + im["n"]]:   # 8-byte records: piece-local offsets
                ua, ub, yk, _ = upd_fields(r)
                assert ua < pm["lsize"] and ub < pm["lsize"] and yk < pm["ysize"]
        srcs = list(plan.asrc[pm["as0"]:pm["as0"] + pm["nas"]])
        srcs += list(plan.usrc[pm["us0"]:pm["us0"] + pm["nus"]])
        for a in srcs:
            q = writer[int(a["uoff"])]
            assert rank[q] < rank[p]
            if p in depth_of:
                assert q in depth_of and depth_of[q] < depth_of[p]
    # factor storage of a piece: blocks sorted by size class (36 | 18 | 10 doubles), the row-in-piece blocks numbered in block order
    for p in range(plan.npiece):
        pm = P[p]
        blks = plan.blk[pm["b0"]:pm["b0"] + pm["nb"]]
        di, dj = blks["info"] & 15, (blks["info"] >> 4) & 15
        size = (di * dj + 1) & ~1
        n36, n18 = int((size == 36).sum()), int((size == 18).sum())
        assert (n36, n18) == (int(pm["n36"]), int(pm["n18"]))
        start = {36: 0, 18: 36 * n36, 10: 36 * n36 + 18 * n18}
        seen = {36: 0, 18: 0, 10: 0}
        rank = 0
        for bm, s in zip(blks, size):
            s = int(s)
            assert int(bm["off"]) - int(pm["lbase"]) == start[s] + s * seen[s]
            seen[s] += 1
            if int(bm["info"]) & (1 << 10):
                info = int(bm["info"])
                assert (((info >> 11) & 31) | (((info >> 24) & 255) << 5)) == rank
                rank += 1
        assert rank == int(pm["nint"]) and int(pm["lsize"]) == 36 * n36 + 18 * n18 + 10 * (len(blks) - n36 - n18)
    # a launch never mixes depths; its LDS reservation covers its pieces; budgets stay below the hardware's 160 KiB
    lds = [int(v) for v in plan.plv_lds_f] + [int(v) for v in plan.plv_lds_b] + [plan.tail_lds_f, plan.tail_lds_b]
    assert max(lds) * 8 <= 158 * 1024


@pytest.mark.parametrize("interleave", [False, True])
def test_plan_small_graph_matches_dense_cholesky(hip_lib, interleave):
    g = make_graph(60, 12, seed=3)
    plan, H, b = _plan_and_system(hip_lib, g, interleave)
    _structure_invariants(plan)
    _check(plan, H, b, 0.0)
    _check(plan, H, b, 7.5)


def test_plan_with_tiny_pieces_exercises_every_phase(hip_lib):
    """Caps far below the defaults force many pieces, external phases, split lists (partial tiles) and a multi-piece tail."""
    g = make_graph(150, 30, seed=5)
    env = {"cap_leaf": 400, "cap_tail": 700, "tail_width": 2, "min_chunk": 1, "pcap_leaf": 16, "nt_leaf": 256}
    plan, H, b = _plan_and_system(hip_lib, g, False, env)
    assert plan.npiece > 20 and len(plan.tail_pieces) >= 2 and len(plan.plv_ptr) > 2
    assert len(plan.mb) > 0 and len(plan.umb) > 0 and np.any(plan.piece["nas"] > 0) and np.any(plan.piece["nus"] > 0)
    _structure_invariants(plan)
    _check(plan, H, b, 1e-3)
    # pieces of equal depth packed into execution groups (one workgroup factors several subtrees side by side)
    plan3, H3, b3 = _plan_and_system(hip_lib, g, False, dict(env, group_cap=1500, nt_leaf=512))
    assert plan3.npiece < plan.npiece
    _structure_invariants(plan3)
    _check(plan3, H3, b3, 1e-3)
    # same system, no tail: every piece goes through the per-depth launches
    plan2, H2, b2 = _plan_and_system(hip_lib, g, False, dict(env, tail_width=0))
    assert len(plan2.tail_pieces) == 0
    _check(plan2, H2, b2, 1e-3)
    # a depth with many pieces is launched in parts, by LDS need (sorted inside the depth; cuts where a CU holds 32 / 24 / 16 workgroups)
    g5 = make_graph(400, 80, seed=2)
    plan5, H5, b5 = _plan_and_system(hip_lib, g5, False, {"tail_width": 2})
    plan4, H4, b4 = _plan_and_system(hip_lib, g5, False, {"tail_width": 2, "split_min": 4})
    assert len(plan4.plv_ptr) > len(plan5.plv_ptr) and plan4.npiece == plan5.npiece
    assert int(plan4.plv_lds_b[0]) < int(plan5.plv_lds_b[0])   # the small pieces of the first depth no longer reserve what its largest needs
    _structure_invariants(plan4)
    _check(plan4, H4, b4, 1e-3)


def test_mid_class_pieces_between_the_bottom_and_the_tail(hip_lib):
    """Round 5: the depths between the bushy bottom and the tail are cut with a larger cap and launched with wider workgroups (one launch
    per depth next to the leaf pieces of that depth); their internal updates come as right-looking lists like the tail's.  Fewer, larger
    pieces there, less update-matrix storage, same factor."""
    g = make_graph(600, 120, seed=3)
    base = {"tail_width": 2, "cap_leaf": 500, "mid_width": 0, "flow": 0}   # (flow = 0: not the single-launch regime, which has no mid class and one workgroup size)
    p0, H0, b0 = _plan_and_system(hip_lib, g, False, base)
    assert not np.any(p0.piece["pad5"] == 1) and not np.any(p0.plv_cls == 1)
    p1, H1, b1 = _plan_and_system(hip_lib, g, False, dict(base, mid_width=12, cap_mid=1600))
    mid = p1.piece["pad5"] == 1
    assert mid.sum() >= 3 and np.any(p1.plv_cls == 1) and np.any(p1.plv_cls == 0) and len(p1.tail_pieces) >= 1
    assert p1.npiece < p0.npiece and p1.unz < p0.unz and p1.lnz == p0.lnz
    # a launch holds pieces of one class, and its workgroup size is that class's
    for l in range(len(p1.plv_ptr) - 1):
        cls = p1.piece["pad5"][p1.plv_pieces[p1.plv_ptr[l]:p1.plv_ptr[l + 1]]]
        assert np.all(cls == p1.plv_cls[l]) and p1.plv_nt[l] == (128 if p1.plv_cls[l] else 64)   # (mid pieces: 128 threads since round 6)
    assert p1.piece["nc"][mid].mean() > p0.piece["nc"][p0.piece["pad5"] == 0].mean()
    _structure_invariants(p1)
    _check(p1, H1, b1, 1e-3)
    # no tail at all: the mid class runs up to the root
    p2, H2, b2 = _plan_and_system(hip_lib, g, False, dict(base, mid_width=12, cap_mid=1600, tail_width=0))
    assert len(p2.tail_pieces) == 0 and np.any(p2.piece["pad5"] == 1)
    _structure_invariants(p2)
    _check(p2, H2, b2, 1e-3)


def test_multi_graph_batch_plan_with_groups_mid_and_tail(hip_lib):
    """The throughput plan of round 5 on a BATCH (groups of leaf pieces on 128-thread workgroups, mid pieces, a tail per graph; the
    defaults of batches >= 32, forced here onto six small graphs): the executor factors the block-diagonal system of all graphs; a group
    never mixes graphs, a launch never mixes classes."""
    from semantic_slam_amd import GraphSLAM
    gs = [make_graph(40 + 5 * k, 8 + k, seed=70 + k) for k in range(6)]
    gps = [GraphProblem.from_synth(g, interleave=bool(k & 1)) for k, g in enumerate(gs)]
    os.environ["SSLAM_CHOL_OPTS"] = "nt_leaf=128,group_cap=600,cap_leaf=150,mid_width=3,cap_mid=400,tail_width=1,front=1"
    try:
        plan = Plan(hip_lib, [GraphSLAM.from_problem(gp) for gp in gps])
    finally:
        os.environ.pop("SSLAM_CHOL_OPTS")
    cls = plan.piece["pad5"]
    assert (cls == 0).sum() >= 6 and (cls == 1).sum() >= 6 and (cls == 2).sum() >= 6 and set(plan.plv_nt) == {128}   # (leaf groups and, since round 6, mid pieces: 128 threads)
    for p in range(plan.npiece):     # every column of a piece (group) belongs to the piece's graph
        pm = plan.piece[p]
        assert np.all(plan.col["graph"][pm["c0"]:pm["c0"] + pm["nc"]] == pm["graph"])
    _structure_invariants(plan)
    # block-diagonal system in the batch's internal row order: the pose rows of all graphs, then the landmark rows of all graphs
    Hs, bs, npose = [], [], []
    for gp in gps:
        U, b = gp.linearize()
        Hg = (U + U.T).toarray() - np.diag(U.diagonal())
        idx, n = _internal_order(gp)
        Hs.append(Hg[np.ix_(idx, idx)]); bs.append(b[idx])
        h, _ = gp.hessian_index()
        npose.append(6 * sum(1 for v in range(gp.nv) if gp.vtype[v] == 0 and h[v] >= 0))
    dim = sum(len(b) for b in bs)
    assert dim == plan.dim
    P0 = sum(npose)
    pos_p, pos_l, where = 0, P0, []
    for b, npz in zip(bs, npose):
        where.append(np.concatenate([np.arange(pos_p, pos_p + npz), np.arange(pos_l, pos_l + len(b) - npz)]))
        pos_p += npz; pos_l += len(b) - npz
    H = np.zeros((dim, dim)); bb = np.zeros(dim)
    for Hg, b, w in zip(Hs, bs, where):
        H[np.ix_(w, w)] = Hg; bb[w] = b
    _check(plan, H, bb, 1e-3)


def test_front_tables_are_a_fraction_of_the_record_tables(hip_lib):
    """Round 6: the front tables of the throughput plan (batch regime forced onto one 600-pose graph): every piece has its blob, the blobs tile
    fblob in 16-byte pieces, the tables are a fraction of the record plan's, the LDS a workgroup needs does not grow, and a plan whose
    pieces the packed tables cannot hold (more than two columns per 8-lane team of a level) reports why and keeps the record kernels."""
    g = make_graph(600, 120, seed=11)
    env = {"nt_leaf": 128, "group_cap": 2800, "cap_leaf": 700, "mid_width": 60, "tail_width": 6, "order_mul": 1.5, "order_add": 2, "ustage": 0, "flow": 0}
    plan, H, b = _plan_and_system(hip_lib, g, True, env)
    assert plan.front and len(plan.fgrp) == plan.npiece
    fg = plan.fgrp
    order = np.argsort(fg["blob"])
    assert fg["blob"][order[0]] == 0 and np.all(fg["blob"][order][1:] == (fg["blob"] + fg["words"])[order][:-1]) and np.all(fg["words"] % 4 == 0)
    assert int((fg["blob"] + fg["words"]).max()) == len(plan.fblob) and np.all(fg["graph"] == plan.piece["graph"])
    record_bytes = sum(a.nbytes for a in (plan.blk, plan.col, plan.upd, plan.item, plan.uitem, plan.asrc, plan.usrc, plan.ilv, plan.mb, plan.umb, plan.rupd, plan.piece))
    assert 4 * len(plan.fblob) < 0.35 * record_bytes
    assert np.all(plan.plv_lds_ff <= plan.plv_lds_f + 128)      # LDS doubles per launch: residency is LDS-bound
    _check(plan, H, b, 1e-3)
    G = plan.front_group(int(plan.plv_pieces[0]))
    assert G["ncomp"] >= 2 and G["nchild"] == 0                  # a group of leaf pieces of the bottom depth: several components, no children
    assert any(plan.front_group(p)["n2"] > 0 for p in range(plan.npiece))     # blocks with a second child source exist (the rank lists are exercised)


def test_parent_links_and_both_orderings_factor(hip_lib):
    """Every piece names the piece its update matrix goes to (what the dependency-driven launch k_chol_flow waits on); both elimination
    orders (multiple minimum degree over independent sets, round 4; lowest-index minimum degree) give a valid plan, and the new one a
    shallower tree."""
    g = make_graph(150, 30, seed=5)
    plan, H, b = _plan_and_system(hip_lib, g, False, {"cap_leaf": 400, "tail_width": 2})
    _structure_invariants(plan)
    _check(plan, H, b, 1e-3)
    order_of = {int(p): k for k, p in enumerate(list(plan.plv_pieces) + list(plan.tail_pieces))}
    par = plan.piece["pad4"]
    assert (par >= -1).all() and (par == -1).sum() >= 1
    for p, q in enumerate(par):
        if q >= 0:
            assert order_of[int(q)] > order_of[p]
    g2 = make_graph(600, 120, seed=3)
    lv = {}
    for order in ("mmd", "mindeg"):
        p2, H2, b2 = _plan_and_system(hip_lib, g2, False, {"order": order})
        _structure_invariants(p2)
        _check(p2, H2, b2, 1e-3)
        lv[order] = p2.nlevels
    assert lv["mmd"] < lv["mindeg"]


def test_plan_plane_landmarks_and_S_config(hip_lib):
    g = make_graph(500, 100, seed=1, landmark_kind="plane")
    plan, H, b = _plan_and_system(hip_lib, g, True)
    _structure_invariants(plan)
    _check(plan, H, b, 1e-2)


def test_bitset_ordering_gives_the_plan_of_the_list_ordering(hip_lib):
    """Graphs of <= 2048 nodes (what the orchestrator re-plans at every tick) are ordered on adjacency bitsets (chol_plan.hpp
    multi_min_degree_bits); SSLAM_CHOL_OPTS order_bits_max=0 keeps the sorted-list form that large graphs use.  Same candidates, same tie breaks:
    every array of the plan is identical."""
    names = ("col", "blk", "upd", "item", "mb", "ilv", "piece", "asrc", "usrc", "fwd", "uitem", "umb", "rcol", "rupd")
    for (n, m, seed, kind) in [(110, 39, 4, "point"), (436, 149, 9, "point"), (300, 60, 2, "plane"), (37, 5, 1, "point")]:
        g = make_graph(n, m, seed=seed, landmark_kind=kind)
        a, _, _ = _plan_and_system(hip_lib, g, False)
        b, _, _ = _plan_and_system(hip_lib, g, False, {"order_bits_max": 0})
        assert a.ncol == b.ncol and a.lnz == b.lnz and a.unz == b.unz and a.nlevels == b.nlevels
        for name in names:
            x, y = getattr(a, name), getattr(b, name)
            assert x.shape == y.shape and x.tobytes() == y.tobytes(), name
        assert list(a.plv_pieces) == list(b.plv_pieces) and list(a.tail_pieces) == list(b.tail_pieces)


def test_plans_of_the_growing_orchestrator_graph_factor(hip_lib):
    """The structures a tick of the orchestrator really hands to the symbolic phase: the keyframe chain with its landmark edges as it
    grows over a replay (oracle/oracle_slam.c drives the growth: semantic_graph_slam.cpp:104-150), planned in the small-batch regime the
    tick runs in (one graph: per-piece workgroups, no groups) -- tiny graphs with a handful of columns included."""
    from oracle.oracle import SlamTickC
    from semantic_slam_amd import GraphSLAM
    from semantic_slam_amd.synth import make_replay
    events, _ = make_replay(7, n_samples=260, n_landmarks=24)
    o = SlamTickC(const_stddev_x=0.00667, const_stddev_q=0.00001)
    ticks, checked = 0, 0
    for ev in events:
        if ev.objects is not None:
            o.set_segmented_objects(ev.objects)
        o.vio(ev.stamp[0], ev.stamp[1], ev.odom)
        if not (ev.run_after and o.run()):
            continue
        ticks += 1
        if ticks not in (1, 2, 5, 12, 25, 40):
            continue
        g = o.graph()
        vfixed = np.zeros(len(g["vtype"]), np.int32); vfixed[0] = 1          # graph_slam.cpp:109-111
        gp = GraphProblem(g["vtype"], vfixed, g["est"], g["etype"], g["evi"], g["evj"], g["meas"], g["info"])
        plan = Plan(hip_lib, [GraphSLAM.from_problem(gp)])
        _structure_invariants(plan)
        U, b = gp.linearize()
        Hg = (U + U.T).toarray() - np.diag(U.diagonal())
        idx, n = _internal_order(gp)
        assert n == plan.dim
        _check(plan, Hg[np.ix_(idx, idx)], b[idx], 1e-4 * np.abs(Hg).max())
        checked += 1
    assert checked >= 4
