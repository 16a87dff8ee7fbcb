"""Optional cross-check against a real g2o ("if the GPU box happens to have g2o"): never required, not part of the tests.
usage: python tools/crosscheck_g2o.py <graph.g2o> [iterations]
Runs the `g2o` command-line tool (if one is on PATH) with the reference's algorithm ("lm_var", graph_slam.cpp:67-73) on a graph saved
by sslam_graph_save_g2o / GraphSLAM.save, runs this library on the same file, and prints the largest difference of the optimised
vertex estimates.  Without a g2o binary it says so and exits 0."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_vertices(path):
    out = {}
    for line in open(path):
        t = line.split()
        if t and t[0] in ("VERTEX_SE3:QUAT", "VERTEX_TRACKXYZ"):
            out[int(t[1])] = np.array([float(v) for v in t[2:]])
    return out


def main():
    if len(sys.argv) < 2:
        print(__doc__); return 2
    exe = shutil.which("g2o")
    if exe is None:
        print("no g2o binary on PATH: nothing to cross-check against (this tool is optional)")
        return 0
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    src = sys.argv[1]
    out = os.path.join(tempfile.mkdtemp(), "g2o_out.g2o")
    subprocess.check_call([exe, "-i", str(iters), "-solver", "lm_var", "-o", out, src])
    ref = read_vertices(out)
    from semantic_slam_amd import GraphSLAM
    G = GraphSLAM(); G.load(src); G.optimize(iters)
    worst = 0.0
    for vid, est in ref.items():
        mine = np.asarray(G.estimate(vid))[:len(est)]
        if len(est) == 7 and np.dot(mine[3:], est[3:]) < 0:
            mine = np.concatenate([mine[:3], -mine[3:]])
        worst = max(worst, float(np.abs(mine - est).max()))
    print(f"{len(ref)} vertices, largest estimate difference vs g2o after {iters} iterations: {worst:.3e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
