"""TEST INFRASTRUCTURE — CPU restatement (NumPy, float64) of the point-to-plane ICP of row J1 (sslam_seg_icp_point_to_plane).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  The reference tree holds no ICP
(BASELINE.json north_star names one); parity unpinned: this states the textbook Gauss-Newton point-to-plane step
(Chen & Medioni 1992; Low 2004 for the linearisation) independently of the HIP code:

    minimise  sum_i (n_k(i) . (T p_i) + d_k(i))^2,   T <- (exp[w]x, u) o T,   J_i = [ (q_i x n)^T  n^T ],  q_i = T p_i
"""
import numpy as np


def _exp(w):
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th == 0:
        return np.eye(3)
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K)


def icp_point_to_plane(xyz, labels, planes, iterations, T0=None):
    xyz = np.asarray(xyz, np.float32).astype(np.float64).reshape(-1, 3)
    labels = np.asarray(labels, np.int64)
    planes = np.asarray(planes, np.float32).astype(np.float64).reshape(-1, 4)
    ok = (labels >= 0) & (labels < len(planes)) & np.isfinite(xyz).all(1)
    P, N, D = xyz[ok], planes[labels[ok], :3], planes[labels[ok], 3]
    R, t = (np.eye(3), np.zeros(3)) if T0 is None else (np.asarray(T0[:9], float).reshape(3, 3), np.asarray(T0[9:], float))
    for _ in range(iterations):
        if len(P) < 6:
            break
        Q = P @ R.T + t
        r = (N * Q).sum(1) + D
        J = np.hstack([np.cross(Q, N), N])
        dx = np.linalg.solve(J.T @ J, -(J.T @ r))
        E = _exp(dx[:3])
        R, t = E @ R, E @ t + dx[3:]
    Q = P @ R.T + t
    r = (N * Q).sum(1) + D
    rms = float(np.sqrt((r * r).sum() / len(P))) if len(P) else 0.0
    return np.concatenate([R.reshape(9), t]), rms, int(len(P))
