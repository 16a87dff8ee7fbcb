import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from semantic_slam_amd import GraphSLAM, GraphBatch
from semantic_slam_amd.synth import make_graph
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G0 = GraphSLAM.from_synth(make_graph(5000, 1000, seed=0)); p = tempfile.mktemp(suffix=".g2o"); G0.save(p)
gs = []
for k in range(B):
    G = GraphSLAM(); G.load(p)
    gs.append(G)
b = GraphBatch(gs); b.upload()
ms = b.time_linearize(10)
print("ms_per_build", ms, "GB/s", b.linearize_bytes() / ms / 1e6, "bytes", b.linearize_bytes())
