"""TEST INFRASTRUCTURE — an independent NumPy / SciPy restatement of the planar-segmentation frontend.

Second, differently-structured restatement of what oracle_seg.c restates in scalar C (PCL 1.7's
IntegralImageNormalEstimation + OrganizedMultiPlaneSegmentation::segmentAndRefine as configured at reference
src/planar_segmentation/plane_segmentation.cpp:84-106,136-156; algorithm summary in SURVEY.md A.6 / A.7), used only by
tests/test_oracle_seg.py to pin the C oracle the way oracle/np_graph.py pins the backend oracle:

  * integral images by cumulative sums, window sums / covariances / eigen-decomposition vectorised over the whole box,
  * connected components from the comparator's edge masks with scipy.sparse.csgraph (no raster union-find),
  * per-region plane fit with ordered float32 accumulation (numpy.add.accumulate is sequential, like PCL's loop),
  * only the genuinely sequential recurrences (chamfer distance map, the two refinement sweeps, the Moore boundary trace) are
    Python loops.

Same decisions as the C oracle where SURVEY.md and PCL differ: depth-dependent comparator thresholds,
boundary trace from the LAST inlier, float covariance, the three trigonometric calls of pcl::computeRoots in double, no
left neighbour at column 0 in the backward refinement sweep.  PARITY UNPINNED, like everything that restates PCL here.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

F = np.float32


# ---------------------------------------------------------------------------------------------- pcl::eigen33, vectorised
def _roots2(b, c):
    d = b * b - F(4.0) * c
    d = np.where(d < 0, F(0), d).astype(F)
    sd = np.sqrt(d)
    return np.stack([np.zeros_like(b), F(0.5) * (b - sd), F(0.5) * (b + sd)], axis=-1)


def _compute_roots(m):
    """m: [...,3,3] float32 symmetric (scaled).  Returns ascending roots [...,3] (pcl::computeRoots)."""
    m00, m01, m02, m11, m12, m22 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2], m[..., 1, 1], m[..., 1, 2], m[..., 2, 2]
    c0 = m00 * m11 * m22 + F(2.0) * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01
    c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12
    c2 = m00 + m11 + m22
    quad = _roots2(c2, c1)
    s_inv3 = F(1.0) / F(3.0)
    s_sqrt3 = np.sqrt(F(3.0))
    c2_over_3 = c2 * s_inv3
    a_over_3 = (c1 - c2 * c2_over_3) * s_inv3
    a_over_3 = np.where(a_over_3 > 0, F(0), a_over_3).astype(F)
    half_b = F(0.5) * (c0 + c2_over_3 * (F(2.0) * c2_over_3 * c2_over_3 - c1))
    q = half_b * half_b + a_over_3 * a_over_3 * a_over_3
    q = np.where(q > 0, F(0), q).astype(F)
    rho = np.sqrt(-a_over_3)
    theta = np.arctan2(np.sqrt(-q).astype(np.float64), half_b.astype(np.float64)).astype(F) * s_inv3
    cos_t = np.cos(theta.astype(np.float64)).astype(F)
    sin_t = np.sin(theta.astype(np.float64)).astype(F)
    r0 = c2_over_3 + F(2.0) * rho * cos_t
    r1 = c2_over_3 - rho * (cos_t + s_sqrt3 * sin_t)
    r2 = c2_over_3 - rho * (cos_t - s_sqrt3 * sin_t)
    # the three conditional swaps of pcl::computeRoots
    sw = r0 >= r1
    r0, r1 = np.where(sw, r1, r0), np.where(sw, r0, r1)
    sw = r1 >= r2
    r1n, r2n = np.where(sw, r2, r1), np.where(sw, r1, r2)
    sw2 = sw & (r0 >= r1n)
    r0, r1n = np.where(sw2, r1n, r0), np.where(sw2, r0, r1n)
    cubic = np.stack([r0, r1n, r2n], axis=-1).astype(F)
    use_quad = (np.abs(c0) < F(1.1920929e-07)) | (cubic[..., 0] <= 0)
    return np.where(use_quad[..., None], quad, cubic).astype(F)


def eigen33(mat):
    """Smallest eigenpair of symmetric float32 3x3 matrices [...,3,3] -> (eigenvalue [...], eigenvector [...,3])."""
    mat = np.asarray(mat, F)
    scale = np.abs(mat).reshape(mat.shape[:-2] + (9,)).max(axis=-1)
    scale = np.where(scale <= F(1.17549435e-38), F(1.0), scale).astype(F)
    s = (mat / scale[..., None, None]).astype(F)
    r = _compute_roots(s)
    ev = (r[..., 0] * scale).astype(F)
    s = s.copy()
    for k in range(3):
        s[..., k, k] = s[..., k, k] - r[..., 0]

    def cross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                         a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1).astype(F)
    v1, v2, v3 = cross(s[..., 0, :], s[..., 1, :]), cross(s[..., 0, :], s[..., 2, :]), cross(s[..., 1, :], s[..., 2, :])
    l1 = v1[..., 0] * v1[..., 0] + v1[..., 1] * v1[..., 1] + v1[..., 2] * v1[..., 2]
    l2 = v2[..., 0] * v2[..., 0] + v2[..., 1] * v2[..., 1] + v2[..., 2] * v2[..., 2]
    l3 = v3[..., 0] * v3[..., 0] + v3[..., 1] * v3[..., 1] + v3[..., 2] * v3[..., 2]
    pick1 = (l1 >= l2) & (l1 >= l3)
    pick2 = ~pick1 & (l2 >= l1) & (l2 >= l3)
    v = np.where(pick1[..., None], v1, np.where(pick2[..., None], v2, v3))
    l = np.where(pick1, l1, np.where(pick2, l2, l3)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        vec = (v / np.sqrt(l)[..., None]).astype(F)
    return ev, vec


# ---------------------------------------------------------------------------------------------- normals
def distance_map(depth, factor):
    """depth-change map + two-pass chamfer transform (float32), including PCL's wrap-around read at column 0."""
    h, w = depth.shape
    fin = np.isfinite(depth)
    thr = F(factor) * (np.abs(depth) + F(1.0)) * F(2.0)
    marked = np.zeros((h, w), bool)
    with np.errstate(invalid="ignore"):
        br = (np.abs(depth[:-1, :-1] - depth[:-1, 1:]) > thr[:-1, :-1]) | ~fin[:-1, :-1] | ~fin[:-1, 1:]
        bd = (np.abs(depth[:-1, :-1] - depth[1:, :-1]) > thr[:-1, :-1]) | ~fin[:-1, :-1] | ~fin[1:, :-1]
    marked[:-1, :-1] |= br | bd
    marked[:-1, 1:] |= br
    marked[1:, :-1] |= bd
    dm = np.where(marked, F(0), F(w + h)).astype(F).reshape(-1)
    dm = np.concatenate([dm, np.zeros(1, F)])
    a14, a10 = F(1.4), F(1.0)
    for r in range(1, h):
        p, c0 = (r - 1) * w, r * w
        for c in range(1, w):
            m = min(min(dm[p + c - 1] + a14, dm[p + c] + a10), min(dm[c0 + c - 1] + a10, dm[p + c + 1] + a14))
            if m < dm[c0 + c]:
                dm[c0 + c] = m
    for r in range(h - 2, -1, -1):
        nx, c0 = (r + 1) * w, r * w
        for c in range(w - 2, -1, -1):
            m = min(min(dm[nx + c - 1] + a14, dm[nx + c] + a10), min(dm[c0 + c + 1] + a10, dm[nx + c + 1] + a14))
            if m < dm[c0 + c]:
                dm[c0 + c] = m
    return dm[:-1].reshape(h, w)


def normals(pts, factor=0.03, smoothing_size=20.0):
    """pts [h,w,3] float32 -> (normals [h,w,4] float32 (nx,ny,nz,curvature; NaN where undefined), distance map [h,w])."""
    pts = np.asarray(pts, F)
    h, w, _ = pts.shape
    fin = np.isfinite(pts[..., 0] + pts[..., 1] + pts[..., 2])
    p64 = np.where(fin[..., None], pts, 0).astype(np.float64)
    prods = []
    for a in range(3):
        for b in range(a, 3):
            prods.append(np.where(fin, (pts[..., a] * pts[..., b]).astype(F), F(0)).astype(np.float64))   # float product, double sum
    vals = np.stack([p64[..., k] for k in range(3)] + prods, axis=-1)          # [h, w, 9] what a finite pixel adds
    # PCL's recurrence, in its own order of operations:  S(r+1,c+1) = ((S(r,c+1) + S(r+1,c)) - S(r,c)) + v   (doubles; the nine
    # channels are carried side by side, rows and columns stay sequential)
    ii = np.zeros((h + 1, w + 1, 9))
    cnt = np.zeros((h + 1, w + 1))
    cnt[1:, 1:] = np.cumsum(np.cumsum(fin.astype(np.float64), axis=1), axis=0)   # integers: exact in any order
    for r in range(h):
        prev, cur = ii[r], ii[r + 1]
        acc = np.zeros(9)
        fr, vr = fin[r], vals[r]
        for c in range(w):
            acc = prev[c + 1] + acc - prev[c]
            if fr[c]:
                acc = acc + vr[c]
            cur[c + 1] = acc
    integ = [ii[..., k] for k in range(9)] + [cnt]
    dm = distance_map(pts[..., 2], factor)
    out = np.full((h, w, 4), np.nan, F)
    border = int(smoothing_size)
    rr, cc = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    smooth = np.minimum(dm, F(smoothing_size))
    ok = (rr >= border) & (rr < h - border) & (cc >= border) & (cc < w - border) & np.isfinite(pts[..., 2]) & (smooth > F(2.0))
    if not ok.any():
        return out, dm
    r, c = rr[ok], cc[ok]
    s = smooth[ok].astype(np.int64)
    sx, sy = c - s // 2, r - s // 2

    def win(ii):
        return ii[sy + s, sx + s] + ii[sy, sx] - ii[sy, sx + s] - ii[sy + s, sx]
    count = win(integ[9])
    keep = count > 0
    cen = np.stack([win(integ[k]).astype(F) for k in range(3)], axis=-1)
    so = [win(integ[3 + k]).astype(F) for k in range(6)]
    cov = np.empty(cen.shape[:-1] + (3, 3), F)
    cov[..., 0, 0], cov[..., 0, 1], cov[..., 0, 2] = so[0], so[1], so[2]
    cov[..., 1, 0], cov[..., 1, 1], cov[..., 1, 2] = so[1], so[3], so[4]
    cov[..., 2, 0], cov[..., 2, 1], cov[..., 2, 2] = so[2], so[4], so[5]
    fc = count.astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        for a in range(3):
            for b in range(3):
                cov[..., a, b] = cov[..., a, b] - (cen[..., a] * cen[..., b]) / fc
        ev, v = eigen33(cov)
        p = pts[r, c]
        ct = (F(0) - p[..., 0]) * v[..., 0] + (F(0) - p[..., 1]) * v[..., 1] + (F(0) - p[..., 2]) * v[..., 2]
        v = np.where((ct < 0)[..., None], -v, v)
        curv = np.where(ev > 0, np.abs(ev / (cov[..., 0, 0] + cov[..., 1, 1] + cov[..., 2, 2])), F(0)).astype(F)
    res = np.concatenate([v, curv[..., None]], axis=-1).astype(F)
    res[~keep] = np.nan
    out[r, c] = res
    return out, dm


# ---------------------------------------------------------------------------------------------- multi-plane segmentation
def multi_plane(pts, nrm, min_inliers=500, angular_threshold=0.017453 * 2, distance_threshold=0.02, maximum_curvature=0.001,
                max_regions=64):
    """Returns (regions: list of dict(model, centroid, inliers, last_inlier), labels [h,w] int32 (-1 = no plane),
    cc_labels [h,w] int32 (-1 invalid), contours: list of index arrays)."""
    pts = np.asarray(pts, F); nrm = np.asarray(nrm, F)
    h, w, _ = pts.shape
    n = h * w
    P = pts.reshape(n, 3); N = nrm.reshape(n, 4)
    ang_thr = F(np.cos(np.float64(F(angular_threshold))))      # cosf(angular_threshold)
    pd = (P[:, 0] * N[:, 0] + P[:, 1] * N[:, 1] + P[:, 2] * N[:, 2]).astype(F)
    finite = np.isfinite(P[:, 0])
    idx = np.arange(n).reshape(h, w)

    def cmp(i1, i2):   # PlaneCoefficientComparator::compare(i1, i2), depth dependent on i1
        z = P[i1, 2]
        thr = F(distance_threshold) * (z * z)
        with np.errstate(invalid="ignore"):
            dot = N[i1, 0] * N[i2, 0] + N[i1, 1] * N[i2, 1] + N[i1, 2] * N[i2, 2]
            return (np.abs(pd[i1] - pd[i2]) < thr) & (dot > ang_thr)
    left_i, left_j = idx[:, 1:].ravel(), idx[:, :-1].ravel()
    up_i, up_j = idx[1:, :].ravel(), idx[:-1, :].ravel()
    el = cmp(left_i, left_j) & finite[left_i]
    eu = cmp(up_i, up_j) & finite[up_i]
    # the raster algorithm only ever links labelled pixels: a link to a non-finite neighbour cannot pass the comparator (NaN)
    rows = np.concatenate([left_i[el], up_i[eu]]); cols = np.concatenate([left_j[el], up_j[eu]])
    g = sp.coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    # compact ids in order of the first pixel of every component (PCL: order of the root's provisional id), finite pixels only
    cc = np.full(n, -1, np.int64)
    fin_idx = np.nonzero(finite)[0]
    first_of = {}
    nxt = 0
    comp_f = comp[fin_idx]
    order = np.full(comp.max() + 1, -1, np.int64)
    for ci in comp_f:            # first occurrence order
        if order[ci] < 0:
            order[ci] = nxt; nxt += 1
    cc[fin_idx] = order[comp_f]
    counts = np.bincount(cc[cc >= 0], minlength=nxt)
    regions, label_to_model, grow = [], {}, np.zeros(nxt + 1, bool)
    for l in range(nxt):
        if not (counts[l] > min_inliers):
            continue
        m = np.nonzero(cc == l)[0]
        x, y, z = P[m, 0], P[m, 1], P[m, 2]
        seq = lambda a: np.add.accumulate(a.astype(F), dtype=F)[-1]
        acc = np.array([seq(x * x), seq(x * y), seq(x * z), seq(y * y), seq(y * z), seq(z * z), seq(x), seq(y), seq(z)], F)
        a = (acc / F(counts[l])).astype(F)
        cov = np.empty((3, 3), F)
        cov[0, 0] = a[0] - a[6] * a[6]; cov[0, 1] = a[1] - a[6] * a[7]; cov[0, 2] = a[2] - a[6] * a[8]
        cov[1, 1] = a[3] - a[7] * a[7]; cov[1, 2] = a[4] - a[7] * a[8]; cov[2, 2] = a[5] - a[8] * a[8]
        cov[1, 0], cov[2, 0], cov[2, 1] = cov[0, 1], cov[0, 2], cov[1, 2]
        ev, v = eigen33(cov)
        p = np.array([v[0], v[1], v[2], 0], F)
        p[3] = F(-1) * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8])
        ct = (F(0) - a[6]) * p[0] + (F(0) - a[7]) * p[1] + (F(0) - a[8]) * p[2]
        if ct < 0:
            p[:3] = -p[:3]
            p[3] = F(-1) * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8])
        es = cov[0, 0] + cov[1, 1] + cov[2, 2]
        curv = abs(ev / es) if es != 0 else F(0)
        if curv < F(maximum_curvature) and len(regions) < max_regions:
            label_to_model[l] = len(regions); grow[l] = True
            regions.append(dict(model=p.copy(), centroid=a[6:9].copy(), inliers=int(counts[l]), last_inlier=int(m[-1]), label=l))
    # refine(): two sweeps of the PlaneRefinementComparator (sequential by construction)
    lab = cc.copy()
    dthr = F(distance_threshold)

    def rcompare(i1, i2):
        cl, nl = lab[i1], lab[i2]
        if not grow[cl] or grow[nl]:
            return False
        mdl = regions[label_to_model[cl]]["model"]
        d = abs(float(mdl[0] * P[i2, 0] + mdl[1] * P[i2, 1] + mdl[2] * P[i2, 2] + mdl[3]))
        z = P[i1, 2]
        return d < float(dthr * (z * z))

    def take(i2, cl):
        lab[i2] = cl
        R = regions[label_to_model[cl]]
        R["inliers"] += 1; R["last_inlier"] = int(i2)
    for r in range(h - 1):
        cur, nx = r * w, (r + 1) * w
        for c in range(w - 1):
            cl = lab[cur + c]
            if cl < 0 or lab[cur + c + 1] < 0:
                continue
            if rcompare(cur + c, cur + c + 1):
                take(cur + c + 1, cl)
            if lab[nx + c] < 0:
                continue
            if rcompare(cur + c, nx + c):
                take(nx + c, cl)
    for r in range(h - 1, 0, -1):
        cur, prv = r * w, (r - 1) * w
        for c in range(w - 1, -1, -1):
            cl = lab[cur + c]
            if cl < 0:
                continue
            if c >= 1:
                if lab[cur + c - 1] < 0:
                    continue
                if rcompare(cur + c, cur + c - 1):
                    take(cur + c - 1, cl)
            if lab[prv + c] < 0:
                continue
            if rcompare(cur + c, prv + c):
                take(prv + c, cl)
    labels = np.array([label_to_model[l] if (l >= 0 and grow[l]) else -1 for l in lab], np.int32).reshape(h, w)
    # Moore boundary trace from the last inlier of every region
    dxs = (-1, -1, 0, 1, 1, 1, 0, -1); dys = (0, -1, -1, -1, 0, 1, 1, 1)
    contours = []
    for R in regions:
        start = R["last_inlier"]; label = lab[start]
        cx, cy, ci = start % w, start // w, start
        dirn = -1
        for d in range(8):
            x, y = cx + dxs[d], cy + dys[d]
            if 0 <= x < w and 0 <= y < h and lab[ci + dys[d] * w + dxs[d]] != label:
                dirn = d; break
        pts_c = []
        if dirn != -1:
            pts_c.append(start)
            guard = 4 * n + 8
            while True:
                nI = 0
                for d in range(1, 9):
                    nI = (dirn + d) & 7
                    x, y = cx + dxs[nI], cy + dys[nI]
                    if 0 <= x < w and 0 <= y < h and lab[ci + dys[nI] * w + dxs[nI]] == label:
                        break
                dirn = (nI + 4) & 7
                ci += dys[nI] * w + dxs[nI]; cx += dxs[nI]; cy += dys[nI]
                pts_c.append(ci)
                guard -= 1
                if ci == start or guard <= 0:
                    break
        contours.append(np.array(pts_c, np.int32))
    return regions, labels, cc.reshape(h, w).astype(np.int32), contours


def polygon_area(P, idx):
    """pcl::calculatePolygonArea with float accumulation in contour order."""
    P = np.asarray(P, F).reshape(-1, 3)
    res = np.zeros(3, F)
    n = len(idx)
    for i in range(n):
        a, b = P[idx[i]], P[idx[(i + 1) % n]]
        res = (res + np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)).astype(F)
    return F(np.sqrt(res[0] * res[0] + res[1] * res[1] + res[2] * res[2]) * F(0.5))
