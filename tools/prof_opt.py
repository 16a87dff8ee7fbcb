"""Profiling helper: a batch of B copies of the 5000/1000 graph, optimise K LM iterations (run under rocprofv3).
usage: python tools/prof_opt.py <B> <K>"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from semantic_slam_amd import GraphSLAM, GraphBatch
from semantic_slam_amd.synth import make_graph
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tmp = tempfile.mkdtemp()
paths = []
for d in range(min(4, B)):
    G0 = GraphSLAM.from_synth(make_graph(5000, 1000, seed=d)); p = os.path.join(tmp, f"g{d}.g2o"); G0.save(p); paths.append(p)
gs = []
for k in range(B):
    G = GraphSLAM(); G.load(paths[k % len(paths)]); gs.append(G)
b = GraphBatch(gs); b.upload()
st = b.optimize(K)
print("iterations", st[0].iterations, "chi2", st[0].chi2_after, "seconds", st[0].seconds)
