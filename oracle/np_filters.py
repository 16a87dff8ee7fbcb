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
