#!/usr/bin/env python
"""Regenerates the committed golden fixtures.

The reference (/root/reference) has no tests, golden vectors or runnable build (ROS + g2o + PCL),
so these vectors are produced by this repository's own CPU oracle (oracle/oracle_graph.c,
oracle/oracle_seg.c) and cross-checked at generation time against the independent numpy/scipy
restatement (oracle/np_graph.py).  They pin the oracle against silent drift and give the `-m gpu`
tests a fixed target that does not depend on the generator code path.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from semantic_slam_amd.synth import make_graph, make_frame  # noqa: E402
from oracle.oracle import GraphProblem  # noqa: E402
from oracle.np_graph import NpGraph  # noqa: E402
from oracle import oracle  # noqa: E402


def write_g2o(path, gp):
    with open(path, "w") as f:
        f.write("PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1\n")
        for v in range(gp.nv):
            e = gp.est[v]
            if gp.vtype[v] == 0:
                f.write("VERTEX_SE3:QUAT %d %s\n" % (v, " ".join(repr(float(x)) for x in e[:7])))
            elif gp.vtype[v] == 1:
                f.write("VERTEX_TRACKXYZ %d %s\n" % (v, " ".join(repr(float(x)) for x in e[:3])))
            else:
                f.write("VERTEX_PLANE %d %s\n" % (v, " ".join(repr(float(x)) for x in e[:4])))
            if gp.vfixed[v]:
                f.write("FIX %d\n" % v)
        for k in range(gp.ne):
            z, W = gp.meas[k], gp.info[k]
            if gp.etype[k] == 0:
                up = [W[r * 6 + c] for r in range(6) for c in range(r, 6)]
                f.write("EDGE_SE3:QUAT %d %d %s %s\n" % (gp.evi[k], gp.evj[k], " ".join(repr(float(x)) for x in z[:7]), " ".join(repr(float(x)) for x in up)))
            else:
                up = [W[r * 3 + c] for r in range(3) for c in range(r, 3)]
                if gp.etype[k] == 1:
                    f.write("EDGE_SE3_TRACKXYZ %d %d 0 %s %s\n" % (gp.evi[k], gp.evj[k], " ".join(repr(float(x)) for x in z[:3]), " ".join(repr(float(x)) for x in up)))
                else:
                    f.write("EDGE_SE3_PLANE %d %d %s %s\n" % (gp.evi[k], gp.evj[k], " ".join(repr(float(x)) for x in z[:4]), " ".join(repr(float(x)) for x in up)))


def main():
    # ---- backend: 20 poses / 5 landmarks, points and planes ------------------------------------
    for kind in ("point", "plane"):
        g = make_graph(20, 5, seed=11, landmark_kind=kind)
        gp = GraphProblem.from_synth(g, interleave=True)
        write_g2o(os.path.join(HERE, f"graph20_{kind}.g2o"), gp)
        U, b = gp.linearize()
        chi0 = gp.chi2()
        st = gp.optimize(25)
        if kind == "point":  # cross-check with the independent numpy restatement
            G = NpGraph(g); G.optimize(25)
            assert abs(G.chi2() - st.chi2_after) < 1e-9 * st.chi2_after
        np.savez(os.path.join(HERE, f"graph20_{kind}_expected.npz"), chi2_before=chi0, chi2_after=st.chi2_after,
                 estimates=gp.est, b=b, H_upper_data=U.data, H_upper_indices=U.indices, H_upper_indptr=U.indptr)
    # ---- frontend: one 64x48 crop is too small for the 20 px border; use a 96x72 patch ------------
    lib = oracle.lib()
    pts = None
    for seed in range(21, 200):   # the first seed whose patch holds at least two accepted planes and a drop-out hole
        f = make_frame(seed=seed, n_boxes=1, box_w=96, box_h=72)
        b = f.boxes[0]
        cand = np.ascontiguousarray(f.xyz()[b["tl_y"]:b["tl_y"] + 72, b["tl_x"]:b["tl_x"] + 96].reshape(-1, 3))
        nrm0 = np.zeros((96 * 72, 4), np.float32)
        lib.os_normals(cand.ctypes.data_as(C.c_void_p), 96, 72, C.c_float(0.03), C.c_float(20.0), nrm0.ctypes.data_as(C.c_void_p), None)
        regs0 = (C.c_byte * (64 * 48))(); lab0 = np.zeros(96 * 72, np.int32); cont0 = np.zeros(4 * 96 * 72 + 16, np.int32); cptr0 = np.zeros(65, np.int32)
        n0 = lib.os_multi_plane(cand.ctypes.data_as(C.c_void_p), nrm0.ctypes.data_as(C.c_void_p), 96, 72, C.c_uint(100), C.c_float(0.017453 * 2),
                                C.c_float(0.02), C.c_float(0.001), regs0, 64, lab0.ctypes.data_as(C.c_void_p), None,
                                cont0.ctypes.data_as(C.c_void_p), cptr0.ctypes.data_as(C.c_void_p), len(cont0))
        if n0 >= 2 and np.isnan(cand[:, 2]).any():
            pts = cand
            print("golden patch: seed", seed, "regions", n0, "NaN pixels", int(np.isnan(cand[:, 2]).sum()))
            break
    assert pts is not None
    nrm = np.zeros((96 * 72, 4), np.float32)
    dist = np.zeros(96 * 72, np.float32)
    lib.os_normals(pts.ctypes.data_as(C.c_void_p), 96, 72, C.c_float(0.03), C.c_float(20.0), nrm.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))

    class R(C.Structure):
        _fields_ = [("c", C.c_float * 3), ("m", C.c_float * 4), ("inl", C.c_int), ("last", C.c_int), ("first", C.c_int), ("label", C.c_int)]
    regs = (R * 64)(); lab = np.zeros(96 * 72, np.int32); cc = np.zeros(96 * 72, np.int32)
    cont = np.zeros(4 * 96 * 72 + 16, np.int32); cptr = np.zeros(65, np.int32)
    n = lib.os_multi_plane(pts.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p), 96, 72, C.c_uint(100), C.c_float(0.017453 * 2),
                           C.c_float(0.02), C.c_float(0.001), regs, 64, lab.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p),
                           cont.ctypes.data_as(C.c_void_p), cptr.ctypes.data_as(C.c_void_p), len(cont))
    assert n >= 1
    models = np.array([list(regs[k].m) for k in range(n)], np.float32)
    inl = np.array([regs[k].inl for k in range(n)], np.int32)
    np.savez_compressed(os.path.join(HERE, "patch96x72.npz"), points=pts, normals=nrm, distance_map=dist, labels=lab, cc_labels=cc,
                        models=models, inliers=inl, contour_ptr=cptr[:n + 1], contour=cont[:cptr[n]])
    print("golden fixtures written to", HERE)


def make_hull():
    """row a15: RANSAC plane -> projected inliers -> 2-D hull on a noisy plane patch with outliers (oracle/oracle_seg.c);
    the hull vertex set is cross-checked against scipy's qhull at generation time"""
    from scipy.spatial import ConvexHull
    lib = oracle.lib()
    rng = np.random.default_rng(77)
    n = 3000
    nrm = np.array([0.25, -0.35, -1.0]); nrm /= np.linalg.norm(nrm)
    u = np.cross(nrm, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
    ab = rng.normal(0, 0.6, (n, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v + 1.2 * nrm + rng.normal(0, 0.003, (n, 1)) * nrm).astype(np.float32)
    out = rng.choice(n, n // 5, replace=False)
    pts[out] += rng.uniform(-0.4, 0.4, (len(out), 3)).astype(np.float32)
    seed, thr, iters, prob = 5, 0.01, 50, 0.99
    coeff = np.zeros(4, np.float32); inl = np.zeros(n, np.int32)
    k = lib.os_ransac_plane(pts.ctypes.data_as(C.c_void_p), n, C.c_float(thr), iters, C.c_double(prob), C.c_uint64(seed),
                            coeff.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), n, None, None)
    inl = inl[:k].copy()
    proj = np.zeros((k, 3), np.float32)
    lib.os_project_inliers(pts.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), k, coeff.ctypes.data_as(C.c_void_p), proj.ctypes.data_as(C.c_void_p))
    hull = np.zeros(k, np.int32); axes = C.c_int(-9)
    h = lib.os_convex_hull_2d(proj.ctypes.data_as(C.c_void_p), k, hull.ctypes.data_as(C.c_void_p), k, C.byref(axes))
    hull = hull[:h].copy()
    cols = {0: [0, 1], 1: [1, 2], 2: [0, 2]}[axes.value]
    assert set(ConvexHull(proj[:, cols].astype(np.float64)).vertices.tolist()) == set(hull.tolist())
    np.savez_compressed(os.path.join(HERE, "hull3000.npz"), points=pts, seed=seed, threshold=thr, max_iterations=iters, probability=prob,
                        coeff=coeff, inliers=inl, axes=axes.value, hull=hull, hull_points=proj[hull])
    print("hull fixture:", k, "inliers,", h, "hull vertices, axes", axes.value)


def make_tick():
    """rows f2 / f3: a short replay through the C tick driver (oracle/oracle_slam.c), cross-checked tick by tick against the NumPy tick
    (oracle/np_slam.py) in tests/test_oracle_slam.py; what is stored: the graph's structure and the state after the last tick"""
    from semantic_slam_amd.synth import make_replay
    from oracle.oracle import SlamTickC
    events, _ = make_replay(3, n_samples=400, n_landmarks=24)
    c = SlamTickC(const_stddev_x=0.00667, const_stddev_q=0.00001)
    per_tick = []
    for ev in events:
        if ev.objects is not None:
            c.set_segmented_objects(ev.objects)
        c.vio(ev.stamp[0], ev.stamp[1], ev.odom)
        if ev.run_after and c.run():
            st = c.last_stats
            per_tick.append((st.keyframes_added, st.landmarks_added, st.landmarks_matched, st.landmark_edges_added) + tuple(c.counts()))
    g = c.graph(); lm = c.landmarks()
    np.savez_compressed(os.path.join(HERE, "tick400.npz"), seed=3, n_samples=400, n_landmarks=24, per_tick=np.array(per_tick, np.int32),
                        vtype=g["vtype"], etype=g["etype"], evi=g["evi"], evj=g["evj"], est=g["est"], meas=g["meas"],
                        landmark_vertex=lm["vertex"], landmark_class=lm["class_id"], landmark_pose=lm["pose"], landmark_cov=lm["covariance"],
                        robot_pose=c.robot_pose())
    print("tick fixture:", len(per_tick), "ticks,", c.counts())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hull":
        make_hull()
    elif len(sys.argv) > 1 and sys.argv[1] == "tick":
        make_tick()
    else:
        main()
        make_hull()
        make_tick()
