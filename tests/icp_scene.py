"""Synthetic scene for the point-to-plane ICP tests (row J1): points on a floor, two walls and a slanted board, moved by a known
small rigid transform; labels say which plane a point belongs to (-1 = none), some points are NaN like sensor drop-outs."""
import numpy as np


def make_icp_scene(seed=0, n_per_plane=4000, noise=2e-3, angle=0.03, shift=0.05):
    rng = np.random.default_rng(seed)
    planes = np.array([[0, 0, 1, 0.0], [1, 0, 0, -3.0], [0, 1, 0, 2.0], [0.6, 0.0, 0.8, -1.0]], np.float64)
    pts, lab = [], []
    for k, (a, b, c, d) in enumerate(planes):
        n = np.array([a, b, c])
        u = np.cross(n, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(n, u)
        uv = rng.uniform(-2, 2, (n_per_plane, 2))
        p = -d * n + uv[:, :1] * u + uv[:, 1:] * v + rng.normal(0, noise, (n_per_plane, 1)) * n
        pts.append(p); lab.append(np.full(n_per_plane, k))
    pts = np.concatenate(pts); lab = np.concatenate(lab)
    # the cloud as seen after the sensor moved: p_obs = T_true^-1 p, so that ICP has to find T_true
    w = rng.normal(size=3); w *= angle / np.linalg.norm(w)
    th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    t = rng.normal(size=3); t *= shift / np.linalg.norm(t)
    obs = (pts - t) @ R            # R^T (p - t)
    lab = lab.copy()
    lab[rng.random(len(lab)) < 0.05] = -1
    obs[rng.random(len(obs)) < 0.02] = np.nan
    return obs.astype(np.float32), lab.astype(np.int32), planes.astype(np.float32), np.concatenate([R.reshape(9), t])


def small_motion(seed=0, angle=0.02, shift=0.03):
    """a small rigid motion (R, t) for the batched ICP test"""
    rng = np.random.default_rng(seed)
    w = rng.normal(size=3); w *= angle / np.linalg.norm(w)
    th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    t = rng.normal(size=3); t *= shift / np.linalg.norm(t)
    return R, t
