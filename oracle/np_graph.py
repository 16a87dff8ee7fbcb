"""TEST INFRASTRUCTURE — numpy/scipy restatement of the ps_graph_slam backend maths.

Only ``tests/`` (and experiments) may import this.  It is an *independent second
restatement* (vectorised numpy + ``scipy.sparse``) used to cross-check the plain-C oracle
(``oracle/oracle_graph.c``) and the HIP path.  PARITY UNPINNED: the reference
(``/root/reference``) has no tests or golden vectors for this path and g2o is not
vendored/installed, so the arithmetic below restates g2o's published
types from SURVEY.md Appendix A:

* ``VertexSE3`` oplus  X <- X * fromVectorMQT(d)            (g2o types/slam3d, A.4)
* ``EdgeSE3``      e = toVectorMQT(Z^-1 Xi^-1 Xj)             (reference call site graph_slam.cpp:136-148)
* ``EdgeSE3PointXYZ`` e = Ri^T (p - ti) - z                  (graph_slam.cpp:150-166)
* ``EdgeSE3Plane`` e = (Xi^-1 ∘ pi_w) ⊖ pi_meas, numeric J   (include/g2o/edge_se3_plane.hpp:15-24)
* Levenberg-Marquardt accept/reject rule                     (g2o OptimizationAlgorithmLevenberg, A.3;
                                                               driven from graph_slam.cpp:199-205)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def qmul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qconj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def qrot(q, v):
    qv = q[..., :3]
    t = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * t + np.cross(qv, t)


def qmat(q):
    x, y, z, w = np.moveaxis(q, -1, 0)
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def skew(v):
    S = np.zeros(v.shape[:-1] + (3, 3))
    S[..., 0, 1] = -v[..., 2]; S[..., 0, 2] = v[..., 1]
    S[..., 1, 0] = v[..., 2]; S[..., 1, 2] = -v[..., 0]
    S[..., 2, 0] = -v[..., 1]; S[..., 2, 1] = v[..., 0]
    return S


def pose_oplus(X, d):
    """VertexSE3::oplus: X <- X * fromVectorMQT(d), d = [dt, dq_xyz] (A.4)."""
    v = d[..., 3:]
    w2 = 1.0 - np.sum(v * v, axis=-1, keepdims=True)
    ok = w2 >= 0
    dq = np.concatenate([np.where(ok, v, 0.0), np.where(ok, np.sqrt(np.maximum(w2, 0)), 1.0)], axis=-1)
    t = X[..., :3] + qrot(X[..., 3:], d[..., :3])
    q = qmul(X[..., 3:], dq)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return np.concatenate([t, q], axis=-1)


def se3_error_jac(Xi, Xj, Z, want_jac=True):
    """EdgeSE3 error (6) and Jacobians wrt the oplus increments of Xi, Xj."""
    qzi = qconj(Z[..., 3:])
    qii = qconj(Xi[..., 3:])
    tb = qrot(qii, Xj[..., :3] - Xi[..., :3])          # translation of B = Xi^-1 Xj
    qb = qmul(qii, Xj[..., 3:])
    te = qrot(qzi, tb - Z[..., :3])                     # E = Z^-1 B
    qe = qmul(qzi, qb)
    sgn = np.where(qe[..., 3:4] < 0, -1.0, 1.0)
    e = np.concatenate([te, sgn * qe[..., :3]], axis=-1)
    if not want_jac:
        return e
    n = Xi.shape[:-1]
    Ra = qmat(qzi)
    Re = qmat(qe)
    Ji = np.zeros(n + (6, 6)); Jj = np.zeros(n + (6, 6))
    Ji[..., :3, :3] = -Ra
    Ji[..., :3, 3:] = 2.0 * Ra @ skew(tb)
    # d xyz(qa * (1,-v) * qb)/dv = -xyz(qa * (v,0) * qb)
    for k in range(3):
        vk = np.zeros(n + (4,)); vk[..., k] = 1.0
        col = qmul(qmul(qzi, vk), qb)[..., :3]
        Ji[..., 3:, 3 + k] = -sgn * col
    Jj[..., :3, :3] = Re
    Jj[..., 3:, 3:] = sgn[..., None] * (qe[..., 3:4, None] * np.eye(3) + skew(qe[..., :3]))
    return e, Ji, Jj


def point_error_jac(Xi, p, z, want_jac=True):
    qi = qconj(Xi[..., 3:])
    pc = qrot(qi, p - Xi[..., :3])
    e = pc - z
    if not want_jac:
        return e
    n = Xi.shape[:-1]
    Ji = np.zeros(n + (3, 6))
    Ji[..., :, :3] = -np.eye(3)
    Ji[..., :, 3:] = 2.0 * skew(pc)
    Jl = qmat(qi)
    return e, Ji, Jl


# ---- Plane3D (g2o slam3d_addons), SURVEY A.4 --------------------------------------------

def plane_normalize(p):
    n = np.linalg.norm(p[..., :3], axis=-1, keepdims=True)
    return p / n


def plane_azimuth(n):
    return np.arctan2(n[..., 1], n[..., 0])


def plane_elevation(n):
    return np.arctan2(n[..., 2], np.hypot(n[..., 0], n[..., 1]))


def plane_rotation(n):
    """Rz(azimuth) * Ry(-elevation)."""
    a = plane_azimuth(n); el = plane_elevation(n)
    ca, sa = np.cos(a), np.sin(a)
    ce, se = np.cos(-el), np.sin(-el)
    R = np.zeros(n.shape[:-1] + (3, 3))
    # Rz(a) @ Ry(b): [[ca*cb, -sa, ca*sb],[sa*cb, ca, sa*sb],[-sb, 0, cb]]
    R[..., 0, 0] = ca * ce; R[..., 0, 1] = -sa; R[..., 0, 2] = ca * se
    R[..., 1, 0] = sa * ce; R[..., 1, 1] = ca; R[..., 1, 2] = sa * se
    R[..., 2, 0] = -se; R[..., 2, 1] = 0.0; R[..., 2, 2] = ce
    return R


def plane_ominus(a, b):
    """a ⊖ b = (azimuth(m), elevation(m), dist(a) - dist(b)),  m = rotation(n_a)^T n_b, dist = -d."""
    m = np.einsum('...ji,...j->...i', plane_rotation(a[..., :3]), b[..., :3])
    return np.stack([plane_azimuth(m), plane_elevation(m), -a[..., 3] + b[..., 3]], axis=-1)


def plane_oplus(p, v):
    a, el, dd = v[..., 0], v[..., 1], v[..., 2]
    s = np.stack([np.cos(el) * np.cos(a), np.cos(el) * np.sin(a), np.sin(el)], axis=-1)
    n = np.einsum('...ij,...j->...i', plane_rotation(p[..., :3]), s)
    d = -(-p[..., 3] + dd)
    return plane_normalize(np.concatenate([n, d[..., None]], axis=-1))


def plane_to_local(X, pw):
    qi = qconj(X[..., 3:])
    ti = -qrot(qi, X[..., :3])
    n = qrot(qi, pw[..., :3])
    d = pw[..., 3] - np.sum(ti * n, axis=-1)
    return np.concatenate([n, d[..., None]], axis=-1)


def plane_error(X, pw, z):
    return plane_ominus(plane_to_local(X, pw), z)


def plane_error_jac(X, pw, z, want_jac=True):
    e = plane_error(X, pw, z)
    if not want_jac:
        return e
    n = X.shape[:-1]
    delta = 1e-9
    Ji = np.zeros(n + (3, 6)); Jl = np.zeros(n + (3, 3))
    for d in range(6):
        dv = np.zeros(n + (6,)); dv[..., d] = delta
        ep = plane_error(pose_oplus(X, dv), pw, z)
        em = plane_error(pose_oplus(X, -dv), pw, z)
        Ji[..., :, d] = (ep - em) / (2 * delta)
    for d in range(3):
        dv = np.zeros(n + (3,)); dv[..., d] = delta
        ep = plane_error(X, plane_oplus(pw, dv), z)
        em = plane_error(X, plane_oplus(pw, -dv), z)
        Jl[..., :, d] = (ep - em) / (2 * delta)
    return e, Ji, Jl


class NpGraph:
    """State = poses [Np,7], landmarks [Nl,3|4]; vertex 0 (pose 0) fixed (graph_slam.cpp:109-111).

    Hessian ordering used here: poses 1..Np-1 (6 each) then landmarks (3 each)."""

    def __init__(self, g):
        self.g = g
        self.poses = g.poses_init.copy()
        self.lms = g.lms_init.copy()
        self.kind = g.landmark_kind
        self.Np = g.n_poses
        self.Nl = g.n_landmarks
        self.dim = 6 * (self.Np - 1) + 3 * self.Nl

    def chi2(self, poses=None, lms=None):
        poses = self.poses if poses is None else poses
        lms = self.lms if lms is None else lms
        g = self.g
        e = se3_error_jac(poses[g.odom_ij[:, 0]], poses[g.odom_ij[:, 1]], g.odom_z, False)
        c = np.einsum('ei,eij,ej->', e, g.odom_info, e)
        if self.kind == "point":
            el = point_error_jac(poses[g.lm_ij[:, 0]], lms[g.lm_ij[:, 1]], g.lm_z, False)
        else:
            el = plane_error_jac(poses[g.lm_ij[:, 0]], lms[g.lm_ij[:, 1]], g.lm_z, False)
        c += np.einsum('ei,eij,ej->', el, g.lm_info, el)
        return float(c)

    def build(self):
        """H (scipy CSC, full symmetric) and b = -J^T Omega e in the ordering above."""
        g = self.g
        Np, Nl = self.Np, self.Nl
        i, j = g.odom_ij[:, 0], g.odom_ij[:, 1]
        e, Ji, Jj = se3_error_jac(self.poses[i], self.poses[j], g.odom_z)
        W = g.odom_info
        rows, cols, vals = [], [], []
        b = np.zeros(self.dim)

        def add_block(bi, bj, off_i, off_j, M, mask):
            r = off_i[:, None, None] + np.arange(M.shape[1])[None, :, None]
            c = off_j[:, None, None] + np.arange(M.shape[2])[None, None, :]
            r = np.broadcast_to(r, M.shape)[mask]; c = np.broadcast_to(c, M.shape)[mask]
            rows.append(r.ravel()); cols.append(c.ravel()); vals.append(M[mask].ravel())

        oi = 6 * (i.astype(np.int64) - 1); oj = 6 * (j.astype(np.int64) - 1)
        mi = i > 0; mj = j > 0
        JiW = np.einsum('eki,ekl->eil', Ji, W); JjW = np.einsum('eki,ekl->eil', Jj, W)
        add_block(i, i, oi, oi, JiW @ Ji, mi)
        add_block(j, j, oj, oj, JjW @ Jj, mj)
        mij = mi & mj
        add_block(i, j, oi, oj, JiW @ Jj, mij)
        add_block(j, i, oj, oi, JjW @ Ji, mij)
        np.subtract.at(b, (oi[mi, None] + np.arange(6)).ravel(), np.einsum('eil,el->ei', JiW, e)[mi].ravel())
        np.subtract.at(b, (oj[mj, None] + np.arange(6)).ravel(), np.einsum('eil,el->ei', JjW, e)[mj].ravel())

        p, l = g.lm_ij[:, 0], g.lm_ij[:, 1]
        if self.kind == "point":
            e, Jp, Jl = point_error_jac(self.poses[p], self.lms[l], g.lm_z)
        else:
            e, Jp, Jl = plane_error_jac(self.poses[p], self.lms[l], g.lm_z)
        W = g.lm_info
        op = 6 * (p.astype(np.int64) - 1); ol = 6 * (Np - 1) + 3 * l.astype(np.int64)
        mp = p > 0; ml = np.ones_like(mp)
        JpW = np.einsum('eki,ekl->eil', Jp, W); JlW = np.einsum('eki,ekl->eil', Jl, W)
        add_block(p, p, op, op, JpW @ Jp, mp)
        add_block(l, l, ol, ol, JlW @ Jl, ml)
        add_block(p, l, op, ol, JpW @ Jl, mp)
        add_block(l, p, ol, op, JlW @ Jp, mp)
        np.subtract.at(b, (op[mp, None] + np.arange(6)).ravel(), np.einsum('eil,el->ei', JpW, e)[mp].ravel())
        np.subtract.at(b, (ol[:, None] + np.arange(3)).ravel(), np.einsum('eil,el->ei', JlW, e).ravel())
        H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                          shape=(self.dim, self.dim)).tocsc()
        return H, b

    def apply(self, dx, poses=None, lms=None):
        poses = (self.poses if poses is None else poses).copy()
        lms = (self.lms if lms is None else lms).copy()
        Np = self.Np
        poses[1:] = pose_oplus(poses[1:], dx[:6 * (Np - 1)].reshape(Np - 1, 6))
        dl = dx[6 * (Np - 1):].reshape(self.Nl, 3)
        if self.kind == "point":
            lms = lms + dl
        else:
            lms = plane_oplus(lms, dl)
        return poses, lms

    def optimize(self, max_iters=1024, verbose=False):
        """g2o OptimizationAlgorithmLevenberg::solve loop."""
        lam = 0.0; nu = 2.0
        its = 0
        hist = []
        for it in range(max_iters):
            cur = self.chi2()
            H, b = self.build()
            if it == 0:
                lam = 1e-5 * H.diagonal().max()
                nu = 2.0
            rho = 0.0; q = 0
            while True:
                try:
                    dx = spla.spsolve((H + lam * sp.identity(self.dim, format='csc')).tocsc(), b)
                    ok = np.all(np.isfinite(dx))
                except Exception:
                    ok = False
                if ok:
                    P, L = self.apply(dx)
                    tmp = self.chi2(P, L)
                else:
                    tmp = np.inf
                scale = float(np.dot(dx, lam * dx + b)) + 1e-3 if ok else 1.0
                rho = (cur - tmp) / scale
                if rho > 0 and np.isfinite(tmp):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha); nu = 2.0
                    cur = tmp
                    self.poses, self.lms = P, L
                else:
                    lam *= nu; nu *= 2.0
                q += 1
                if not (rho < 0 and q < 10):
                    break
            its = it + 1
            hist.append((cur, lam, q))
            if verbose:
                print(it, cur, lam, q)
            if q == 10 or rho == 0:
                break
        return its, hist
