"""TEST INFRASTRUCTURE — CPU restatement (NumPy) of the cloud filters of the reference's legacy path:

    plane_segmentation::distance_filter      plane_segmentation.cpp:607-629
    downsamplePointcloud -> pcl::VoxelGrid   plane_segmentation.cpp:565-581   [UPSTREAM: pcl/filters/voxel_grid.hpp, applyFilter]
    removeOutliers -> pcl::StatisticalOutlierRemoval  plane_segmentation.cpp:583-605   [UPSTREAM: statistical_outlier_removal.hpp]

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  Parity unpinned (PCL is not installed; the
reference has no tests).  Where PCL's result depends on an unspecified order (VoxelGrid sums the points of a voxel in float in the
order of an unstable std::sort) the restatement fixes one: coordinate sums in 2^-20 fixed point, which no order can change.
"""
import numpy as np

F = np.float32


def distance_filter(xyz, dmin=0.3, dmax=3.0):
    p = np.asarray(xyz, F).reshape(-1, 3)
    with np.errstate(invalid="ignore", over="ignore"):
        d = np.sqrt(((p[:, 0] * p[:, 0]).astype(F) + (p[:, 1] * p[:, 1]).astype(F)).astype(F) + (p[:, 2] * p[:, 2]).astype(F), dtype=F).astype(np.float64)
        keep = (d > dmin) & (d < dmax)
    return np.nonzero(keep)[0].astype(np.int32)


def voxel_grid(xyz, leaf=0.1):
    p = np.asarray(xyz, F).reshape(-1, 3)
    fin = np.isfinite(p).all(1)
    q = p[fin]
    if len(q) == 0:
        return np.zeros((0, 3), F), np.zeros(0, np.int32)
    inv = F(1.0) / F(leaf)
    lo, hi = q.min(0), q.max(0)
    minb = np.floor((lo * inv).astype(F)).astype(np.int64)
    maxb = np.floor((hi * inv).astype(F)).astype(np.int64)
    div = maxb - minb + 1
    ijk = np.floor((q * inv).astype(F)).astype(np.int64) - minb
    cell = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    fixed = np.rint(q.astype(np.float64) * 1048576.0).astype(np.int64)
    cells, inverse, counts = np.unique(cell, return_inverse=True, return_counts=True)
    sums = np.zeros((len(cells), 3), np.int64)
    np.add.at(sums, inverse, fixed)
    cent = ((sums.astype(np.float64) / 1048576.0) / counts[:, None].astype(np.float64)).astype(F)
    return cent, counts.astype(np.int32)


def statistical_outlier_removal(xyz, mean_k=50, stddev_mul=1.0):
    p = np.asarray(xyz, F).reshape(-1, 3)
    n = len(p)
    fin = np.isfinite(p).all(1)
    md = np.full(n, -1.0, F)
    idx = np.nonzero(fin)[0]
    q = p[idx]
    for a in range(0, len(q), 512):      # exact k nearest neighbours by brute force, squared distances in float32 like flann::L2_Simple
        blk = q[a:a + 512]
        dx = (blk[:, None, 0] - q[None, :, 0]).astype(F); dy = (blk[:, None, 1] - q[None, :, 1]).astype(F); dz = (blk[:, None, 2] - q[None, :, 2]).astype(F)
        d2 = ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)
        part = np.sort(np.partition(d2, mean_k, axis=1)[:, :mean_k + 1], axis=1)
        s = np.zeros(len(blk), np.float64)
        for k in range(1, mean_k + 1):
            s += np.sqrt(part[:, k], dtype=F).astype(np.float64)
        md[idx[a:a + 512]] = (s / mean_k).astype(F)
    valid = md >= 0
    v = md[valid].astype(np.float64)
    if len(v) < 2:
        return np.zeros(0, np.int32), md
    total, sq = 0.0, 0.0
    for x in v:                           # sequential double sums, index order (applyFilterIndices)
        total += x; sq += x * x
    mean = total / len(v)
    std = np.sqrt((sq - total * total / len(v)) / (len(v) - 1.0))
    thr = mean + stddev_mul * std
    keep = valid & ~(md.astype(np.float64) > thr)
    return np.nonzero(keep)[0].astype(np.int32), md


# ---- plane_segmentation::computeKmeans -> cv::kmeans (plane_segmentation.cpp:524-535), restated as sslam.h describes --------------
_M64 = (1 << 64) - 1


def _mix(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def kmeans(pts, k, seed=0, attempts=10, max_count=10, eps=0.01):
    """[UPSTREAM: OpenCV cv::kmeans loop] for every attempt: random centres; then (update centres from the last labels, stop when the
    largest squared centre shift <= eps^2 or after max(max_count, 2) rounds, else assign) -- the labels kept are those of the last
    assignment, the centres those updated after it; the attempt with the smallest compactness wins."""
    p = np.asarray(pts, F)
    p = p.reshape(len(p), -1)
    n, dim = p.shape
    lo, hi = p.min(0), p.max(0)
    fixed = np.rint(p.astype(np.float64) * 1048576.0).astype(np.int64)
    best, out = np.inf, None
    for a in range(attempts):
        C = np.zeros((k, dim), F)
        labels, compact, assigned = None, 0.0, False
        max_shift = np.inf
        it = 0
        while True:
            if it == 0:
                for j in range(k):
                    for d in range(dim):
                        u = F(_mix(seed ^ (a << 40) ^ (j << 20) ^ d) >> 40) / F(16777216.0)
                        C[j, d] = F(lo[d] + F(u * F(hi[d] - lo[d])))
            else:
                max_shift = 0.0
                for j in range(k):
                    old = C[j].copy()
                    m = labels == j
                    if m.any():
                        C[j] = ((fixed[m].sum(0).astype(np.float64) / 1048576.0) / float(m.sum())).astype(F)
                    max_shift = max(max_shift, float(((C[j].astype(np.float64) - old.astype(np.float64)) ** 2).sum()))
            it += 1
            if it == max(max_count, 2) or max_shift <= eps * eps:
                break
            d2 = np.zeros((n, k), F)
            for d in range(dim):
                t = (p[:, d, None] - C[None, :, d]).astype(F)
                d2 = (d2 + (t * t).astype(F)).astype(F)
            labels = d2.argmin(1)                       # first minimum: lowest centre wins ties
            compact = float(np.rint(d2[np.arange(n), labels].astype(np.float64) * 1048576.0).astype(np.int64).sum()) / 1048576.0
            assigned = True
        if assigned and compact < best:
            best, out = compact, (labels.astype(np.int32), C.copy())
    return out[0], out[1], best
