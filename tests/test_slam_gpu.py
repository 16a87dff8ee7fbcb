"""GPU parity of the orchestrator tick and of data association: the product (C-ABI sslam_slam_*: keyframe
gate, graph growth with structure rebuild every tick, sslam_graph_optimize, marginals by hessian index, association kernel) against
oracle/np_slam.py driven through the same synthetic run.

Reference: src/ps_graph_slam/semantic_graph_slam.cpp:58-205,234-287; include/ps_graph_slam/data_association.h:75-317;
keyframe_updater.hpp:41-65.  Bars: keyframe decisions, match / new decisions and ids bit-exact; association distances 1e-4
relative (float32 on both sides, identical operation order); estimates 1e-6 absolute (LM run to termination on both sides),
covariances 1e-4 relative to the largest entry."""
import numpy as np
import pytest

from semantic_slam_amd.synth import make_replay
from tests.slam_replay import oracle_instance, product_instance, product_instance_with, feed, planes_of
from oracle import np_slam as S

pytestmark = pytest.mark.gpu


def _compare_state(P, Or, tol_est=1e-6):
    lp, lo = P.getMappedLandmarks(), Or.assoc.landmarks
    assert len(lp) == len(lo)
    for a, b in zip(lp, lo):
        assert (a.id, a.vertex, a.class_id, a.plane_type) == (b["id"], b["vertex"], b["class_id"], b["plane_type"])
        cov = np.array(a.covariance[:]).reshape(3, 3)
        assert np.abs(cov - b["covariance"]).max() <= 1e-4 * max(1e-3, np.abs(b["covariance"]).max())
        if a.vertex >= 0:
            assert np.abs(P.graph_vertex(a.vertex) - Or.est[a.vertex][:3]).max() <= tol_est
    ids, est = P.getKeyframes()
    assert list(ids) == [k["node"] for k in Or.keyframes]
    for v, e in zip(ids, est):
        eo = Or.est[v]
        s = 1.0 if np.dot(e[3:], eo[3:]) >= 0 else -1.0
        assert np.abs(e[:3] - eo[:3]).max() <= tol_est and np.abs(e[3:] - s * eo[3:]).max() <= tol_est
    rp, ro = P.getRobotPose(), S.iso_to_tq(Or.robot_pose)
    assert np.abs(rp[:3] - ro[:3]).max() <= tol_est
    mp, mo = P.getMap2OdomTrans(), S.iso_to_tq(Or.map2odom)
    assert np.abs(mp[:3] - mo[:3]).max() <= 10 * tol_est


def _run_both(events):
    """the product and the oracle through the same events, compared tick by tick; returns (product, oracle, ticks)"""
    P, Or = product_instance(), oracle_instance()
    ticks = 0
    for ev in events:
        kp, rp = feed(P, ev, True)
        ko, ro = feed(Or, ev, False)
        assert kp == ko and rp == ro
        if not rp:
            continue
        ticks += 1
        st, so = P.last_stats, Or.last_stats
        assert (st.keyframes_added, st.landmarks_added, st.landmarks_matched, st.landmark_edges_added, bool(st.optimized), bool(st.marginals_ok)) == \
               (so["keyframes_added"], so["landmarks_added"], so["landmarks_matched"], so["landmark_edges_added"], so["optimized"], so["marginals_ok"])
        assert st.keyframes_added <= 10
        if so["optimized"]:
            assert abs(st.opt.chi2_after - so["opt"].chi2_after) <= 1e-6 * max(1.0, so["opt"].chi2_after)
        _compare_state(P, Or)
    return P, Or, ticks


@pytest.mark.parametrize("seed,kw", [(0, {}), (1, {}), (2, dict(detect_from=0, first_run_at=60, run_every=9))])
def test_replay_matches_oracle_tick_by_tick(gpu_lib, seed, kw):
    """seeds 0/1: the node's loop keeps up (one or two keyframes per tick after an initial stall that queues > 10 keyframes);
    seed 2: detections inside the stall and a slow loop -- association on the stale robot_pose_ the reference would use"""
    events, lms = make_replay(seed, n_samples=300, **kw)
    P, Or, ticks = _run_both(events)
    assert ticks > 20 and len(Or.assoc.landmarks) >= 8
    if not kw:   # the loop kept up: the map is the true one (every landmark within the stale-pose error of a true landmark)
        tr = np.array([p for p, _, _ in lms])
        for l in P.getMappedLandmarks():
            assert np.linalg.norm(tr - P.graph_vertex(l.vertex), axis=1).min() < 0.25


def test_same_frame_twin_detections_through_the_full_tick(gpu_lib):
    """two same-class, same-plane-type detections 5 cm apart in one frame: the second one matches the landmark the first one has just
    created, which has no graph vertex yet when find_matches copies it (data_association.h:309 reads the uninitialised node there).
    The tick must add both landmark edges to the one new vertex, not abort (round-2 ADVICE, high)."""
    events, _ = make_replay(4, n_samples=260)
    twins = 0
    for ev in events:
        if ev.objects:
            t = dict(ev.objects[0])
            t["pose"] = ev.objects[0]["pose"] + np.array([0.05, 0.0, 0.0], np.float32)
            ev.objects.insert(1, t)
            twins += 1
    P, Or, ticks = _run_both(events)
    assert twins > 20 and ticks > 20
    # every landmark edge of the oracle's graph points at a real vertex, and some landmark got two edges from one keyframe
    pairs = [(i, j) for t, i, j in zip(Or.etype, Or.evi, Or.evj) if t == S.O.ET_SE3_POINT]
    assert all(j >= 0 for _, j in pairs) and len(set(pairs)) < len(pairs)
    assert P.num_edges() == len(Or.etype)


def test_keyframe_gate_ignores_the_sign_of_the_odometry_quaternion(gpu_lib):
    """q and -q are one rotation: Eigen::Quaterniond(delta.linear()).w() of the reference's gate (keyframe_updater.hpp:52-54) is >= 0
    whatever the sign of the quaternions the odometry source publishes (round-2 ADVICE, medium)"""
    events, _ = make_replay(5, n_samples=200)
    for k, ev in enumerate(events):
        if k % 2:
            ev.odom = ev.odom.copy()
            ev.odom[3:] = -ev.odom[3:]
    plain, _ = make_replay(5, n_samples=200)
    P, Or, ticks = _run_both(events)
    Pp = product_instance()
    for ev in plain:
        feed(Pp, ev, True)
    assert ticks > 10 and len(P.getKeyframes()[0]) == len(Pp.getKeyframes()[0]) == len(Or.keyframes)


def test_first_tick_takes_at_most_ten_keyframes(gpu_lib):
    events, _ = make_replay(3, n_samples=200, first_run_at=150, detect_from=150)
    P = product_instance()
    queued = 0
    for ev in events[:151]:
        k, ran = feed(P, ev, True)
        queued += k
    assert queued > 20 and ran
    assert P.last_stats.keyframes_added == 10 and len(P.getKeyframes()[0]) == 10
    assert P.run() and P.last_stats.keyframes_added == 10
    # 9 odometry edges only in the first tick: GraphSLAM::optimize refuses (< 10 edges, graph_slam.cpp:184-186) and nothing moves


@pytest.mark.parametrize("quirk", [0, 1])
def test_association_kernel_matches_oracle(gpu_lib, quirk):
    """find_matches alone, on a hand-made map: equal-distance ties, same-frame new landmarks as candidates, type gating, the
    Euclidean variant and the carried distance_min of quirk B5"""
    rng = np.random.default_rng(5)
    for use_eq in (0, 1):
        P = product_instance(reference_quirks=quirk, use_maha_dist=0 if use_eq else 1, use_eq_dist=use_eq)
        D = S.DataAssociation(keep_distance_min=bool(quirk), use_maha_dist=not use_eq, use_eq_dist=bool(use_eq))
        pose = np.array([0.3, -0.2, 0.1, 0.02, -0.03, 0.7], np.float32)
        for frame in range(6):
            objs = []
            for k in range(int(rng.integers(3, 40))):
                p = rng.uniform(-3, 3, 3).astype(np.float32)
                if k % 5 == 4 and objs:
                    p = objs[-1]["pose"].copy()          # an exact repeat: matches the landmark its twin just created
                objs.append(dict(pose=p, normal=rng.normal(size=4).astype(np.float32), class_id=int(rng.integers(1, 4)),
                                 plane_type=int(rng.integers(0, 2))))
            got = P.find_matches(planes_of(objs), pose)
            ref = D.find_matches(objs, pose, np.float32(0.0), lambda l: l["pose"])
            assert [(g.is_new, g.id) for g in got] == [(int(r["is_new"]), r["id"]) for r in ref]
            for g, r in zip(got, ref):
                assert np.array_equal(np.array(g.pose[:]), r["pose"]) and np.array_equal(np.array(g.local_pose[:]), r["local_pose"])
                assert np.array_equal(np.array(g.normal[:]), r["normal"])
                if r["distance"] >= 0:
                    assert abs(g.distance - r["distance"]) <= 1e-4 * max(1.0, abs(r["distance"]))
            pose[:3] += rng.normal(0, 0.2, 3).astype(np.float32)
        assert len(P.getMappedLandmarks()) == len(D.landmarks) > 10


def test_tick_parity_through_the_frontend(gpu_lib):
    """Row f3 with the cloud path: keyframes that carry a real synthetic organised cloud + detection boxes go through
    sslam_slam_set_point_cloud / set_detected_objects -> the tick's batched frontend pass (sslam_seg_segment_batch seen from
    pose_to_vector6(robot_pose)) -> association -> graph growth -> optimise -> marginals, against the oracle composition
    matrix2vector -> oracle_seg.c -> np_slam association (reference semantic_graph_slam.cpp:207-232, ros_utils.hpp:90-106).
    Same bars as the pre-segmented replay."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from semantic_slam_amd.synth import make_frame
    events, _ = make_replay(6, n_samples=420, detect_every=3)
    frames = [make_frame(seed=20 + k, n_boxes=10) for k in range(6)]
    seg = PointCloudSegmentation()
    P = product_instance_with(seg)
    Or = oracle_instance(seg_params=seg.params)
    ticks = kfs = with_planes = 0
    nd = 0
    for ev in events:
        if ev.objects is not None:
            fr = frames[nd % len(frames)]; nd += 1
            P.setPointCloudData(fr); P.setDetectedObjectInfo(fr.boxes)
            Or.set_point_cloud(fr); Or.set_detected_objects(fr.boxes)
        kp = bool(P.VIOCallback(ev.stamp, ev.odom)); ko = bool(Or.vio(ev.stamp[0], ev.stamp[1], ev.odom))
        assert kp == ko
        kfs += kp
        if not ev.run_after:
            continue
        rp, ro = bool(P.run()), bool(Or.run())
        assert rp == ro
        if not rp:
            continue
        ticks += 1
        st, so = P.last_stats, Or.last_stats
        assert (st.keyframes_added, st.landmarks_added, st.landmarks_matched, st.landmark_edges_added, bool(st.optimized), bool(st.marginals_ok)) == \
               (so["keyframes_added"], so["landmarks_added"], so["landmarks_matched"], so["landmark_edges_added"], so["optimized"], so["marginals_ok"])
        with_planes += so["landmark_edges_added"]
        if so["optimized"]:
            assert abs(st.opt.chi2_after - so["opt"].chi2_after) <= 1e-6 * max(1.0, so["opt"].chi2_after)
        _compare_state(P, Or)
    assert ticks >= 20 and kfs >= 20 and with_planes >= 40 and len(Or.assoc.landmarks) >= 3
    assert seg.last_overflow() == (0, 0, 0)


def test_three_robots_on_one_gpu_equal_their_solo_replays(gpu_lib):
    """Round 6: several orchestrator handles on one GPU (persistent launches share the device by a budget; the handles' graphs run the
    plain single-launch solve, `speculative_trials` 0, so that their launches overlap).  Three robots replaying different runs concurrently,
    one host thread each, end with the keyframes, landmarks and estimates -- bit for bit -- that each reaches alone."""
    import threading
    from semantic_slam_amd.synth import make_replay
    runs = [make_replay(seed, n_samples=220, n_landmarks=24)[0] for seed in (3, 5, 8)]

    def replay(events, out, k):
        S = product_instance()
        S.set_graph_option("speculative_trials", 0)
        ticks = 0
        for ev in events:
            _, ran = feed(S, ev, True)
            ticks += int(ran)
        ids, est = S.getKeyframes()
        out[k] = (ticks, ids.copy(), est.copy(), [(l.id, l.vertex, tuple(l.pose)) for l in S.getMappedLandmarks()])

    solo = [None] * len(runs)
    for k, ev in enumerate(runs):
        replay(ev, solo, k)
    assert all(s[0] >= 5 and len(s[1]) >= 10 for s in solo)
    for rep in range(2):
        conc = [None] * len(runs)
        th = [threading.Thread(target=replay, args=(ev, conc, k)) for k, ev in enumerate(runs)]
        for t in th: t.start()
        for t in th: t.join()
        for a, b in zip(solo, conc):
            assert b is not None and a[0] == b[0] and a[3] == b[3]
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
