"""GPU parity of the planar-segmentation frontend: HIP path (through the C-ABI) vs the CPU oracle.

Bar (north_star): label / inlier sets bit-exact; normals and plane parameters are float pipelines
with an identical operation order on both sides, so they are compared for exact equality too."""
import numpy as np
import pytest

from semantic_slam_amd.synth import make_frame

pytestmark = pytest.mark.gpu


def _run_both(frame, params=None):
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from oracle.oracle import segment_frame as _oracle_segment
    seg = PointCloudSegmentation(params=params)
    planes = seg.segmentallPointCloudData(frame.robot_pose, frame.cam_angle, frame.boxes, frame)
    ref, nrm, lab = _oracle_segment(frame, seg.params, want_products=True)
    return seg, planes, ref, nrm, lab


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frame_matches_oracle_bit_exact(gpu_lib, seed):
    f = make_frame(seed=seed)
    seg, planes, ref, nrm, lab = _run_both(f)
    off = 0
    nvalid = 0
    for bi, b in enumerate(f.boxes):
        n = int(b["width"]) * int(b["height"])
        gn = seg.normals(bi).reshape(n, 4)
        on = nrm[off:off + n]
        assert np.array_equal(np.isnan(gn), np.isnan(on)), f"box {bi}: NaN pattern of the normals differs"
        m = ~np.isnan(on[:, 0])
        nvalid += int(m.sum())
        assert np.array_equal(gn[m], on[m]), f"box {bi}: normals differ"
        gl = seg.labels(bi).reshape(n)
        assert np.array_equal(gl, lab[off:off + n]), f"box {bi}: label image differs in {(gl != lab[off:off+n]).sum()} px"
        off += n
    assert nvalid > 10000
    assert len(planes) == len(ref) and len(ref) > 0
    for a, r in zip(planes, ref):
        assert a.box_index == r.box_index and a.inlier_count == r.inlier_count
        assert a.num_points == r.num_points and a.area == r.area
        assert a.plane_type == ("horizontal" if r.plane_type == 0 else "vertical")
        assert np.array_equal(a.pose, np.array(r.centroid_cam, np.float32))
        assert np.array_equal(a.normal_orientation, np.array(r.normal_d, np.float32))
        assert np.array_equal(a.world_pose, np.array(r.world_pose, np.float32))


@pytest.mark.parametrize("box_w,box_h,env", [(160, 102, None), (168, 100, None), (128, 96, "SSLAM_SEG_GLOBAL_CC"), (128, 96, "SSLAM_SEG_WAVEFRONT_REFINE")])
def test_box_sizes_either_side_of_the_lds_limit(gpu_lib, monkeypatch, box_w, box_h, env):
    """connected components and refinement run out of LDS for boxes of <= 16384 pixels (160 x 102 = 16320 takes the full 64 KB) and
    through HBM above (168 x 100); both forms, and the HBM forms forced on small boxes, give the oracle's label images"""
    if env:
        monkeypatch.setenv(env, "1")
    f = make_frame(seed=5, n_boxes=12, box_w=box_w, box_h=box_h)
    seg, planes, ref, nrm, lab = _run_both(f)
    off = 0
    for bi, b in enumerate(f.boxes):
        n = int(b["width"]) * int(b["height"])
        gl = seg.labels(bi).reshape(n)
        assert np.array_equal(gl, lab[off:off + n]), f"box {bi}: label image differs in {(gl != lab[off:off+n]).sum()} px"
        off += n
    assert len(planes) == len(ref) and len(ref) > 0
    for a, r in zip(planes, ref):
        assert (a.box_index, a.inlier_count, a.num_points, a.area) == (r.box_index, r.inlier_count, r.num_points, r.area)
        assert np.array_equal(a.normal_orientation, np.array(r.normal_d, np.float32))


def test_noise_free_planes_are_recovered(gpu_lib):
    """A noise-free piecewise-planar scene: every pixel a plane owns (label image of the HIP path) lies on the reported plane.
    PlaneRefinementComparator admits a pixel within 0.02 z^2 of the model (depth-dependent threshold), the fitted inliers
    of a noise-free plane sit at the accuracy of the float32 covariance fit (more than num_point_seg pixels within 5 mm), none
    exceeds the comparator's own bound."""
    f = make_frame(seed=5, noise=0.0, nan_fraction=0.0, n_holes=0)
    seg, planes, ref, _, _ = _run_both(f)
    assert len(planes) == len(ref) and len(planes) > 0
    xyz = f.xyz()
    per_box = {}
    for p in planes:
        per_box.setdefault(p.box_index, []).append(p)
    checked = 0
    for bi, plist in per_box.items():
        b = f.boxes[bi]
        pts = xyz[b["tl_y"]:b["tl_y"] + b["height"], b["tl_x"]:b["tl_x"] + b["width"]].reshape(-1, 3).astype(np.float64)
        lab = seg.labels(bi).reshape(-1)
        # label k = k-th region segmentAndRefine accepted in this box; the post-filter (contour > 100, area, orientation) may drop
        # some of them, so match every reported plane to the label whose pixels it fits
        for p in plist:
            n4 = p.normal_orientation.astype(np.float64)
            best = None
            for k in range(int(lab.max()) + 1):
                m = lab == k
                if not m.any():
                    continue
                d = np.abs(pts[m] @ n4[:3] + n4[3])
                if best is None or np.median(d) < best[0]:
                    best = (float(np.median(d)), float(d.max()), int(m.sum()), float((0.02 * pts[m][:, 2] ** 2).max()), int((d < 5e-3).sum()))
            assert best is not None
            med, dmax, npx, bound, close = best
            assert npx == p.inlier_count
            # the connected component the plane was fitted to (> num_point_seg pixels) lies on it to float32-fit accuracy; what the
            # refinement sweeps added stays inside the comparator's own bound
            assert close > 500 and dmax <= 1.05 * bound
            checked += 1
    assert checked == len(planes)


def test_salt_and_pepper_nans_match_oracle(gpu_lib):
    """the degenerate sensor model (isolated NaN pixels collapse the smoothing windows around them): still bit-exact"""
    f = make_frame(seed=0, nan_fraction=0.02, noise=3e-5, n_holes=0)
    seg, planes, ref, nrm, lab = _run_both(f)
    assert len(planes) == len(ref)
    off = 0
    for bi, b in enumerate(f.boxes):
        n = int(b["width"]) * int(b["height"])
        assert np.array_equal(seg.labels(bi).reshape(n), lab[off:off + n])
        assert np.array_equal(seg.normals(bi).reshape(n, 4), nrm[off:off + n], equal_nan=True)
        off += n


def test_batched_frames_equal_frame_by_frame(gpu_lib):
    """sslam_seg_segment_batch: the boxes of several frames packed into one pass give exactly the per-frame results
    (plane records bit for bit), including a frame without any accepted box; nothing is truncated."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    frames = [make_frame(seed=s, n_boxes=nb) for s, nb in ((0, 32), (1, 12), (2, 32), (3, 8))]
    frames[3].boxes["class_id"][:] = 0                      # every box of this frame is rejected by the class whitelist
    seg = PointCloudSegmentation()
    single = [seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f) for f in frames]
    seg2 = PointCloudSegmentation()
    batched = seg2.segment_frames(frames)
    assert seg2.last_overflow() == (0, 0, 0)
    assert [len(x) for x in batched] == [len(x) for x in single] and sum(len(x) for x in single) > 5 and len(batched[3]) == 0
    for a_list, b_list in zip(batched, single):
        for a, b in zip(a_list, b_list):
            assert (a.box_index, a.inlier_count, a.num_points, a.area, a.plane_type, a.type) == (b.box_index, b.inlier_count, b.num_points, b.area, b.plane_type, b.type)
            assert np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_orientation, b.normal_orientation) and np.array_equal(a.world_pose, b.world_pose)
    # max_out smaller than the number of planes: the rest is counted, not silently lost
    few = seg2.segment_frames(frames, max_planes=3)
    assert sum(len(x) for x in few) == 3 and seg2.last_overflow()[0] == sum(len(x) for x in single) - 3


def _same_planes(a_list, b_list):
    assert len(a_list) == len(b_list)
    for a, b in zip(a_list, b_list):
        assert (a.box_index, a.inlier_count, a.num_points, a.area, a.plane_type, a.type) == (b.box_index, b.inlier_count, b.num_points, b.area, b.plane_type, b.type)
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_orientation, b.normal_orientation) and np.array_equal(a.world_pose, b.world_pose)


def test_xyz_only_clouds_give_the_same_planes(gpu_lib):
    """The frontend reads x, y, z of a point and nothing else (plane_segmentation.cpp:24-82): a 12-byte (packed) or 16-byte (pcl::PointXYZ)
    cloud through the same entry points -- the C-ABI takes any point_step / field offsets -- gives the plane records of the 32-byte
    registered cloud bit for bit (blocking call, batched call and the pipelined submit / collect), at 3.69 instead of 9.83 MB per frame
    over PCIe."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from semantic_slam_amd.synth import repack_xyz
    frames = [make_frame(seed=s, n_boxes=nb) for s, nb in ((0, 32), (1, 12), (2, 32))]
    seg = PointCloudSegmentation()
    ref = [seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f) for f in frames]
    assert sum(len(x) for x in ref) > 5
    for step in (12, 16):
        slim = [repack_xyz(f, step) for f in frames]
        assert slim[0].cloud.nbytes == step * 640 * 480
        seg2 = PointCloudSegmentation()
        for f, r in zip(slim, ref):
            _same_planes(seg2.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f), r)
        for a, r in zip(PointCloudSegmentation().segment_frames(slim), ref):
            _same_planes(a, r)
        got = list(PointCloudSegmentation().segment_stream([slim, slim]))
        for batch in got:
            for a, r in zip(batch, ref):
                _same_planes(a, r)


def test_many_boxes_per_call_equal_frame_by_frame(gpu_lib):
    """a call with more boxes than the chip has CUs switches the per-box kernels to their high-residency forms (256-thread plane fit,
    24-row refinement bands); the label images and plane records stay those of the one-frame calls"""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    frames = [make_frame(seed=40 + s, n_boxes=32) for s in range(18)]          # 576 boxes in one call
    seg = PointCloudSegmentation()
    single = [seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f) for f in frames[5::-1]][::-1]
    lab_single = [seg.labels(bi) for bi in range(32)]                            # label images of frame 0 (the last one-frame call)
    seg2 = PointCloudSegmentation()
    batched = seg2.segment_frames(frames)
    assert seg2.last_overflow() == (0, 0, 0)
    for a_list, b_list in zip(batched[:6], single):
        _same_planes(a_list, b_list)
    assert sum(len(x) for x in batched) > 40
    for bi in range(32):                                                        # the parity hooks address frame 0 of a call
        assert np.array_equal(seg2.labels(bi), lab_single[bi])


def test_pipelined_batches_equal_blocking_calls(gpu_lib):
    """sslam_seg_submit_batch / _collect_batch: two batches in flight on the handle's two pipelines, results in submission order and
    equal to the blocking call; a third submit and a blocking call while batches are in flight are refused"""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from semantic_slam_amd.graph_slam import SslamError
    batches = [[make_frame(seed=60 + 4 * k + j, n_boxes=8 + 4 * j) for j in range(3)] for k in range(4)]
    ref = PointCloudSegmentation()
    want = [ref.segment_frames(b) for b in batches]
    seg = PointCloudSegmentation()
    got = list(seg.segment_stream(batches))
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        for a_list, b_list in zip(g, w_):
            _same_planes(a_list, b_list)
    seg.submit_frames(batches[0]); seg.submit_frames(batches[1])
    with pytest.raises(SslamError):
        seg.submit_frames(batches[2])
    with pytest.raises(SslamError):
        seg.segment_frames(batches[2])
    first = seg.collect_frames(); second = seg.collect_frames()
    for a_list, b_list in zip(first + second, want[0] + want[1]):
        _same_planes(a_list, b_list)
    with pytest.raises(SslamError):
        seg.collect_frames()
    _same_planes(seg.segment_frames(batches[3])[0], want[3][0])              # the handle is usable again


def test_ragged_and_rejected_boxes(gpu_lib):
    """Class filter (point_cloud_segmentation.h:126-130), out-of-bounds crop (plane_segmentation.cpp:34-38),
    too few points (:93-95), an empty box list, mixed box sizes."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from oracle.oracle import segment_frame as _oracle_segment
    f = make_frame(seed=3, n_boxes=8)
    f.boxes["class_id"][0] = 0                                  # not whitelisted
    f.boxes["tl_x"][1] = 600; f.boxes["width"][1] = 128         # crosses the right border -> spurious
    f.boxes["width"][2] = 40; f.boxes["height"][2] = 40          # 1600 px < norm_point_thres
    f.boxes["width"][3] = 200; f.boxes["height"][3] = 150; f.boxes["tl_x"][3] = 100; f.boxes["tl_y"][3] = 200
    f.boxes["width"][4] = 97; f.boxes["height"][4] = 131; f.boxes["tl_x"][4] = 301; f.boxes["tl_y"][4] = 17
    f.boxes["height"][5] = -3
    seg = PointCloudSegmentation()
    planes = seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f)
    ref, nrm, lab = _oracle_segment(f, seg.params, want_products=True)
    assert len(planes) == len(ref)
    for a, r in zip(planes, ref):
        assert a.box_index == r.box_index and a.inlier_count == r.inlier_count and a.area == r.area
        assert a.box_index not in (0, 1, 2, 5)
    off = 0
    for bi in (3, 4, 6, 7):
        n = int(f.boxes[bi]["width"]) * int(f.boxes[bi]["height"])
        assert np.array_equal(seg.labels(bi).reshape(n), lab[off:off + n])
        off += n
    assert seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes[:0], f) == []


def test_full_frame_box(gpu_lib):
    """Maximum box size the reference accepts: the whole 640x480 frame (plane_segmentation.cpp:34-38)."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from oracle.oracle import segment_frame as _oracle_segment
    f = make_frame(seed=4, n_boxes=1)
    f.boxes["tl_x"][0] = 0; f.boxes["tl_y"][0] = 0; f.boxes["width"][0] = 640; f.boxes["height"][0] = 480
    seg = PointCloudSegmentation()
    planes = seg.segmentallPointCloudData(f.robot_pose, f.cam_angle, f.boxes, f)
    ref, nrm, lab = _oracle_segment(f, seg.params, want_products=True)
    assert np.array_equal(seg.labels(0).reshape(-1), lab)
    assert len(planes) == len(ref)
    for a, r in zip(planes, ref):
        assert a.inlier_count == r.inlier_count and a.num_points == r.num_points and a.area == r.area


def test_transform_matches_oracle(gpu_lib):
    import ctypes as C
    from oracle import oracle
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    seg = PointCloudSegmentation()
    pose = np.array([0.3, -1.2, 0.9, 0.05, -0.1, 2.1], np.float32)
    T = seg.transform(pose, 0.59)
    ref = np.zeros(16, np.float32)
    oracle.lib().os_transform_normals_to_world(pose.ctypes.data_as(C.c_void_p), C.c_float(0.59), 1, ref.ctypes.data_as(C.c_void_p))
    assert np.array_equal(T.reshape(-1), ref)


def _oracle_ransac(pts, thr, iters, prob, seed):
    import ctypes as C
    from oracle import oracle
    lib = oracle.lib()
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    coeff = np.zeros(4, np.float32); inl = np.zeros(max(len(pts), 1), np.int32); bi = C.c_int(0)
    k = lib.os_ransac_plane(pts.ctypes.data_as(C.c_void_p), len(pts), C.c_float(thr), iters, C.c_double(prob), C.c_uint64(seed),
                            coeff.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), len(inl), None, C.byref(bi))
    return coeff, inl[:k].copy()


@pytest.mark.parametrize("seed", [0, 7, 42])
def test_ransac_plane_inliers_bit_exact(gpu_lib, seed):
    """SURVEY §8 a15 (plane_segmentation.cpp:639-647): inlier index set and refined coefficients vs the oracle."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    rng = np.random.default_rng(seed)
    n = 20000
    pts = np.zeros((n, 3), np.float32)
    pts[:, :2] = rng.uniform(-1, 1, (n, 2))
    pts[:, 2] = 0.3 * pts[:, 0] - 0.2 * pts[:, 1] + 1.5 + rng.normal(0, 0.003, n)
    out = rng.choice(n, n // 3, replace=False)
    pts[out] += rng.uniform(-0.5, 0.5, (len(out), 3)).astype(np.float32)
    seg = PointCloudSegmentation()
    coeff, inl = seg.ransac_plane(pts, 0.01, 50, 0.99, seed)
    rc, ri = _oracle_ransac(pts, 0.01, 50, 0.99, seed)
    assert np.array_equal(inl, ri) and len(inl) > n // 2
    assert np.array_equal(coeff, rc)
    nrm = np.array([0.3, -0.2, -1.0]); nrm /= np.linalg.norm(nrm)
    assert abs(abs(coeff[:3] @ nrm) - 1) < 1e-4


def test_ransac_plane_edge_cases(gpu_lib):
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    seg = PointCloudSegmentation()
    coeff, inl = seg.ransac_plane(np.zeros((2, 3), np.float32))              # fewer than 3 points
    assert len(inl) == 0 and not coeff.any()
    line = np.stack([np.linspace(0, 1, 50), np.linspace(0, 2, 50), np.linspace(0, 3, 50)], 1).astype(np.float32)
    c1, i1 = seg.ransac_plane(line, 0.01, 5, 0.99, 3)                        # collinear: every sample is rejected
    c2, i2 = _oracle_ransac(line, 0.01, 5, 0.99, 3)
    assert np.array_equal(i1, i2) and np.array_equal(c1, c2)
    f = make_frame(seed=1, n_boxes=1)                                         # a real crop with NaN points
    b = f.boxes[0]
    crop = f.xyz()[b["tl_y"]:b["tl_y"] + b["height"], b["tl_x"]:b["tl_x"] + b["width"]].reshape(-1, 3)
    c1, i1 = seg.ransac_plane(crop, 0.01, 50, 0.99, 11)
    c2, i2 = _oracle_ransac(crop, 0.01, 50, 0.99, 11)
    assert np.array_equal(i1, i2) and np.array_equal(c1, c2, equal_nan=True) and len(i1) > 1000


def _oracle_hull(pts, inl, coeff):
    import ctypes as C
    from oracle import oracle
    lib = oracle.lib()
    pts = np.ascontiguousarray(pts, np.float32); inl = np.ascontiguousarray(inl, np.int32); coeff = np.ascontiguousarray(coeff, np.float32)
    proj = np.zeros((len(inl), 3), np.float32)
    lib.os_project_inliers(pts.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), len(inl), coeff.ctypes.data_as(C.c_void_p),
                           proj.ctypes.data_as(C.c_void_p))
    hull = np.zeros(len(inl), np.int32); axes = C.c_int(-9)
    h = lib.os_convex_hull_2d(proj.ctypes.data_as(C.c_void_p), len(inl), hull.ctypes.data_as(C.c_void_p), len(hull), C.byref(axes))
    return proj, hull[:max(h, 0)].copy(), axes.value


@pytest.mark.gpu
@pytest.mark.parametrize("seed,normal", [(0, (0.3, -0.2, -1.0)), (5, (1.0, 0.05, 0.1)), (9, (0.1, 1.0, -0.05))])
def test_project_inliers_and_convex_hull_bit_exact(gpu_lib, seed, normal):
    """SURVEY §8 a15 (plane_segmentation.cpp:648-662): projected points (float, bit-exact), coordinate-plane choice and hull
    vertex sequence vs the oracle, for planes facing z, x and y (xy / yz / xz projections)."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    rng = np.random.default_rng(seed)
    n = 30000
    nrm = np.array(normal, np.float64); nrm /= np.linalg.norm(nrm)
    # two in-plane directions
    u = np.cross(nrm, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
    ab = rng.normal(0, 1, (n, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v + 1.5 * nrm + rng.normal(0, 0.003, (n, 1)) * nrm).astype(np.float32)
    out = rng.choice(n, n // 4, replace=False)
    pts[out] += rng.uniform(-0.5, 0.5, (len(out), 3)).astype(np.float32)
    seg = PointCloudSegmentation()
    coeff, inl = seg.ransac_plane(pts, 0.01, 50, 0.99, seed)
    assert len(inl) > n // 2
    proj, hull, axes = seg.convex_hull_2d(pts, inl, coeff)
    rproj, rhull, raxes = _oracle_hull(pts, inl, coeff)
    assert axes == raxes == {2: 0, 0: 1, 1: 2}[int(np.argmax(np.abs(nrm)))]
    assert np.array_equal(proj, rproj)
    assert np.array_equal(hull, rhull) and 3 <= len(hull) < 200
    assert np.abs(proj @ coeff[:3] + coeff[3]).max() < 1e-5                 # the projected points lie on the plane
    hp = seg.compute2DConvexHull(pts, seed)                                  # the composed reference entry point
    assert np.array_equal(hp, proj[hull])


@pytest.mark.gpu
def test_convex_hull_many_candidates_and_duplicates(gpu_lib):
    """more candidates than one workgroup holds (hull-of-hulls rounds), duplicated points, collinear input"""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    rng = np.random.default_rng(3)
    n = 60000
    ang = rng.uniform(0, 2 * np.pi, n); rad = rng.uniform(0.93, 1.0, n)          # annulus: the octagon filter keeps most points
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(n)], 1).astype(np.float32)
    pts[1000:2000] = pts[0:1000]                                                  # exact duplicates
    coeff = np.array([0, 0, 1, 0], np.float32)
    inl = np.arange(n, dtype=np.int32)
    seg = PointCloudSegmentation()
    proj, hull, axes = seg.convex_hull_2d(pts, inl, coeff)
    rproj, rhull, raxes = _oracle_hull(pts, inl, coeff)
    assert axes == raxes == 0 and np.array_equal(proj, rproj)
    assert np.array_equal(hull, rhull) and len(hull) > 100
    line = np.stack([np.linspace(0, 1, 50), np.linspace(0, 2, 50), np.zeros(50)], 1).astype(np.float32)
    with pytest.raises(RuntimeError):
        seg.convex_hull_2d(line, np.arange(50, dtype=np.int32), coeff)            # collinear: no 2-D hull


@pytest.mark.gpu
def test_golden_hull_fixture(gpu_lib):
    """tests/golden/hull3000.npz (made by tests/golden/make_golden.py from the oracle, vertex set checked against qhull)"""
    import os
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hull3000.npz"))
    seg = PointCloudSegmentation()
    coeff, inl = seg.ransac_plane(g["points"], float(g["threshold"]), int(g["max_iterations"]), float(g["probability"]), int(g["seed"]))
    assert np.array_equal(inl, g["inliers"]) and np.array_equal(coeff, g["coeff"])
    proj, hull, axes = seg.convex_hull_2d(g["points"], inl, coeff)
    assert axes == int(g["axes"]) and np.array_equal(hull, g["hull"]) and np.array_equal(proj[hull], g["hull_points"])


def test_icp_point_to_plane_matches_oracle(gpu_lib):
    """row J1: Gauss-Newton point-to-plane ICP, one reduction pass per round on the device.  Same transform as the NumPy oracle to
    1e-9 (double sums in a different order), the known motion recovered, residual at the noise level; a plane set that leaves a
    direction free is refused."""
    from oracle.np_icp import icp_point_to_plane
    from tests.icp_scene import make_icp_scene
    from semantic_slam_amd import SslamError
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    seg = PointCloudSegmentation()
    for seed in (0, 1):
        obs, lab, planes, T_true = make_icp_scene(seed=seed)
        for iters in (1, 8):
            T, rms, n = seg.icp_point_to_plane(obs, lab, planes, iters)
            To, rmso, no = icp_point_to_plane(obs, lab, planes, iters)
            assert n == no
            assert np.abs(T - To).max() <= 1e-9 and abs(rms - rmso) <= 1e-9
        assert np.abs(T - T_true).max() < 2e-3 and rms < 3e-3
        # warm start from the answer: nothing moves
        T2, rms2, _ = seg.icp_point_to_plane(obs, lab, planes, 2, T0=T)
        assert np.abs(T2 - T).max() < 1e-6
    obs, lab, planes, _ = make_icp_scene(seed=2)
    only_floor = np.where(lab == 0, 0, -1).astype(np.int32)
    with pytest.raises(SslamError) as ei:
        seg.icp_point_to_plane(obs, only_floor, planes, 3)
    assert ei.value.code == -4


def test_cloud_filters_match_oracle(gpu_lib):
    """row f4: range filter, voxel grid and statistical outlier removal on the xyz of a synthetic frame (NaN drop-outs included):
    index sets and voxel counts bit-exact, centroids and mean neighbour distances bit-exact (fixed-point sums / exact kNN in float32)."""
    from oracle import np_filters as NF
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    seg = PointCloudSegmentation()
    f = make_frame(seed=3)
    xyz = f.xyz().reshape(-1, 3)
    # 1. range filter on the full frame
    keep = seg.distance_filter(xyz)
    assert np.array_equal(keep, NF.distance_filter(xyz)) and 0 < len(keep) < len(xyz)
    assert np.array_equal(seg.distance_filter(xyz, 1.0, 2.0), NF.distance_filter(xyz, 1.0, 2.0))
    # 2. voxel grid (leaf 0.1 like upstream, and a finer one)
    for leaf in (0.1, 0.04):
        cent, cnt = seg.downsamplePointcloud(xyz, leaf)
        co, no = NF.voxel_grid(xyz, leaf)
        assert np.array_equal(cnt, no) and cnt.sum() == np.isfinite(xyz).all(1).sum()
        assert np.array_equal(cent, co)
    # 3. statistical outlier removal on the downsampled cloud (what preprocessPointCloud feeds it) plus a few stray points and a NaN
    cent, _ = seg.downsamplePointcloud(xyz, 0.1)
    stray = np.array([[0.0, 0.0, 0.2], [2.0, -2.0, 0.5], [np.nan, 0, 1]], np.float32)
    cloud = np.vstack([cent, stray])
    for k, mul in ((50, 1.0), (8, 2.0)):
        keep, md = seg.removeOutliers(cloud, k, mul)
        ko, mo = NF.statistical_outlier_removal(cloud, k, mul)
        assert np.array_equal(md, mo)
        assert np.array_equal(keep, ko) and 0 < len(keep) < len(cloud)
    assert len(cloud) - 1 not in keep
    with pytest.raises(Exception):
        seg.removeOutliers(cloud[:10], 50, 1.0)


def test_kmeans_and_legacy_cluster_path_match_oracle(gpu_lib):
    """row f4: computeKmeans (assignment on the GPU, fixed-point centre updates) bit-exact vs the oracle on 3-D normals and 1-D
    distances; clusterAndSegmentAllPlanes composed from it finds the two horizontal surfaces of a synthetic scene and equals the same
    composition over the oracle's k-means."""
    from oracle import np_filters as NF
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    seg = PointCloudSegmentation()
    rng = np.random.default_rng(4)
    cent = np.array([[0, 0, 1.0], [1, 0, 0], [0, -1, 0], [0.6, 0, 0.8]], np.float32)
    nrm = (cent[rng.integers(0, 4, 20000)] + rng.normal(0, 0.04, (20000, 3))).astype(np.float32)
    for seed in (0, 5):
        lab, C, comp = seg.computeKmeans(nrm, 4, seed)
        lo, Co, compo = NF.kmeans(nrm, 4, seed)
        assert np.array_equal(lab, lo) and np.array_equal(C, Co) and comp == compo
    d = np.concatenate([rng.normal(-1.0, 0.01, 5000), rng.normal(-2.5, 0.02, 7000)]).astype(np.float32).reshape(-1, 1)
    lab, C, comp = seg.computeKmeans(d, 2, 3)
    lo, Co, compo = NF.kmeans(d, 2, 3)
    assert np.array_equal(lab, lo) and np.array_equal(C, Co) and comp == compo
    # scene in the camera frame of an identity transformation: a floor patch (z = -1) and a table top (z = -0.4), normals +z, plus two
    # vertical walls; the horizontal filter keeps the +z centre, the distance k-means splits floor and table
    def patch(n, origin, u, v, normal):
        ab = rng.uniform(0, 1, (n, 2))
        p = origin + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, 0.002, (n, 1)) * normal
        return p.astype(np.float32), (normal + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    parts = [patch(6000, np.array([-1, -1, -1.0]), np.array([2.0, 0, 0]), np.array([0, 2.0, 0]), np.array([0, 0, 1.0])),
             patch(4000, np.array([0.2, 0.2, -0.4]), np.array([0.8, 0, 0]), np.array([0, 0.6, 0]), np.array([0, 0, 1.0])),
             patch(5000, np.array([1.5, -1, -1.0]), np.array([0, 2.0, 0]), np.array([0, 0, 1.5]), np.array([-1.0, 0, 0])),
             patch(5000, np.array([-1, 1.5, -1.0]), np.array([2.0, 0, 0]), np.array([0, 0, 1.5]), np.array([0, -1.0, 0]))]
    xyz = np.vstack([p for p, _ in parts]); nr = np.vstack([n for _, n in parts])
    nr[::97] = np.nan
    T = np.eye(4, dtype=np.float32)
    rows = seg.clusterAndSegmentAllPlanes(xyz, nr, T, seed=1)
    rows_o = seg.clusterAndSegmentAllPlanes(xyz, nr, T, seed=1, kmeans=NF.kmeans)
    assert np.array_equal(rows, rows_o) and len(rows) >= 6
    dists = sorted(set(np.round(rows[:, 6].astype(np.float64), 1).tolist()))
    assert np.allclose(dists, [0.4, 1.0])                                 # -(n . p): the table top at z = -0.4 and the floor at z = -1
    assert np.all(np.abs(rows[:, 3:6] - [0, 0, 1]) < 0.05)


def test_ransac_and_icp_over_the_boxes_of_a_batch_match_the_oracle(gpu_lib):
    """BASELINE.json configs[3] (640x480 cloud, 32 boxes of 128x96 per frame, "RANSAC+ICP plane extraction"): sslam_seg_ransac_boxes on the
    crops the segmentation left on the device -- refined coefficients, inlier counts, consumed hypotheses and the inlier index sets of every
    box BIT-EXACT against oracle_seg.c's os_ransac_plane run on the same crop with the same per-box seed -- then sslam_seg_icp_boxes per
    frame over those inliers against NumPy's point-to-plane ICP (1e-9: double sums in another order)."""
    from semantic_slam_amd.segmentation import PointCloudSegmentation
    from oracle.np_icp import icp_point_to_plane
    frames = [make_frame(seed=s) for s in (0, 1)]
    seg = PointCloudSegmentation()
    seg.segment_frames(frames)
    seed = 12345
    recs, ms = seg.ransac_boxes(0.01, 50, 0.99, seed)
    assert len(recs) == 64 and ms > 0
    GOLD = 0x9E3779B97F4A7C15
    crops, inl_ref = [], []
    nonzero = 0
    for q, r in enumerate(recs):
        f = frames[r.frame]; b = f.boxes[r.box_index]
        crop = np.ascontiguousarray(f.xyz()[b["tl_y"]:b["tl_y"] + b["height"], b["tl_x"]:b["tl_x"] + b["width"]].reshape(-1, 3))
        assert r.points == len(crop) == 128 * 96 and r.frame == q // 32
        rc, ri = _oracle_ransac(crop, 0.01, 50, 0.99, (seed + q * GOLD) % (1 << 64))
        assert np.array_equal(np.array(r.coeff[:], np.float32), rc), (q, r.coeff[:], rc)
        assert r.inliers == len(ri)
        if q % 5 == 0 or q in (31, 32, 63):                                  # the index sets of a sample of boxes (a D2H copy each)
            assert np.array_equal(seg.ransac_box_inliers(q), ri)
        nonzero += r.inliers > 500
        crops.append(crop); inl_ref.append(ri)
    assert nonzero >= 48                                                       # the scene is piecewise planar: most boxes hold a plane
    # ICP: every box measures its own RANSAC plane moved by a small known motion -> the motion comes back; identical to the NumPy oracle
    from tests.icp_scene import small_motion
    Rm, tm = small_motion(3)
    planes, box_plane = [], []
    for q, r in enumerate(recs):
        n = np.array(r.coeff[:3], np.float64); d = float(r.coeff[3])
        if r.inliers > 500 and abs(np.linalg.norm(n) - 1) < 1e-3:
            n2 = Rm @ n; planes.append(np.concatenate([n2, [d - n2 @ tm]])); box_plane.append(len(planes) - 1)   # plane of the moved points R p + t
        else:
            box_plane.append(-1)
    planes = np.array(planes, np.float32)
    res, ms2 = seg.icp_boxes(box_plane, planes, 6)
    assert len(res) == 2 and ms2 > 0
    for f in range(2):
        pts = np.concatenate([crops[q][inl_ref[q]] for q in range(32 * f, 32 * f + 32)])
        lab = np.concatenate([np.full(len(inl_ref[q]), box_plane[q], np.int32) for q in range(32 * f, 32 * f + 32)])
        To, rmso, no = icp_point_to_plane(pts, lab, planes, 6)
        T = np.array(res[f].T[:])
        assert res[f].status == 0 and res[f].used == no
        assert np.abs(T - To).max() <= 1e-9 and abs(res[f].rms - rmso) <= 1e-9
        assert np.abs(T[:9].reshape(3, 3) - Rm).max() < 5e-3 and np.abs(T[9:] - tm).max() < 2e-2
    # no plane constrains anything: refused per frame, not a hang; a batch without RANSAC flags is refused
    res3, _ = seg.icp_boxes([0] + [-1] * 63, planes, 3)
    assert res3[0].status == -4 and res3[1].used == 0
    seg.segment_frames(frames[:1])
    from semantic_slam_amd import SslamError
    with pytest.raises(SslamError):
        seg.icp_boxes([-1] * 32, planes, 1)


def test_ransac_boxes_edge_cases(gpu_lib):
    """boxes larger than the LDS staging limit (points read through L2), a box of NaNs only, a tiny batch"""
    from semantic_slam_amd.segmentation import PointCloudSegmentation, default_params
    from semantic_slam_amd.synth import BOX_DTYPE
    f = make_frame(seed=4, n_boxes=3, box_w=200, box_h=150)                    # 30,000 points per box: above the ~12,400 that fit in LDS
    p = default_params(); p.norm_point_thres = 100
    seg = PointCloudSegmentation(params=p)
    seg.segment_frames([f])
    recs, _ = seg.ransac_boxes(0.01, 50, 0.99, 7)
    assert len(recs) == 3
    for q, r in enumerate(recs):
        b = f.boxes[r.box_index]
        crop = np.ascontiguousarray(f.xyz()[b["tl_y"]:b["tl_y"] + b["height"], b["tl_x"]:b["tl_x"] + b["width"]].reshape(-1, 3))
        rc, ri = _oracle_ransac(crop, 0.01, 50, 0.99, (7 + q * 0x9E3779B97F4A7C15) % (1 << 64))
        assert np.array_equal(np.array(r.coeff[:], np.float32), rc) and r.inliers == len(ri)
        assert np.array_equal(seg.ransac_box_inliers(q), ri)
    g = make_frame(seed=5, n_boxes=2, box_w=40, box_h=30)
    xyz = g.xyz(); b0 = g.boxes[0]
    xyz[b0["tl_y"]:b0["tl_y"] + b0["height"], b0["tl_x"]:b0["tl_x"] + b0["width"]] = np.nan     # writes through to the frame's cloud
    seg.segment_frames([g])
    recs, _ = seg.ransac_boxes(0.01, 20, 0.99, 1)
    assert len(recs) == 2 and recs[0].inliers == 0 and not any(recs[0].coeff[:]) and recs[0].best_iteration == -1
    assert len(seg.ransac_box_inliers(0)) == 0
