"""per-kernel times of one batched frontend call (32 frames x 32 boxes): python tools/frontend_kernels.py   (run under rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from semantic_slam_amd.segmentation import PointCloudSegmentation
from semantic_slam_amd.synth import make_frame
fs = [make_frame(seed=s) for s in range(4)]
seg = PointCloudSegmentation()
bf = [fs[k % 4] for k in range(32)]
for _ in range(4):
    seg.segment_frames(bf)
print(seg.last_timing())
# the RANSAC + ICP pass over the boxes of the resident batch (BASELINE.json configs[3]), as bench.py's frontend.ransac_icp leg runs it
import numpy as np
for _ in range(3):
    recs, ms_r = seg.ransac_boxes(0.01, 50, 0.99, 2024)
    planes_r = np.array([list(r.coeff) for r in recs], np.float32)
    ok = np.array([r.inliers > 500 and abs(float(np.linalg.norm(planes_r[q, :3])) - 1.0) < 1e-3 for q, r in enumerate(recs)])
    box_plane = np.where(ok, np.arange(len(recs)), -1).astype(np.int32)
    icp, ms_i = seg.icp_boxes(box_plane, planes_r, 5)
print("ransac ms", ms_r, "icp ms", ms_i)
