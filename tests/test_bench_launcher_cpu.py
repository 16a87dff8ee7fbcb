"""`python bench.py --gpus N` must start N ranks by itself (BASELINE.json configs[4]): the launcher logic, and
the re-execution under torch.distributed.run end to end on CPU (gloo rendezvous on 127.0.0.1, no GPU work: SSLAM_BENCH_LAUNCH_PROBE)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_only_when_ranks_are_missing():
    assert bench.launcher_command(["--gpus", "1"], 1, {}) is None
    assert bench.launcher_command(["--gpus", "8"], 8, {"WORLD_SIZE": "8"}) is None       # the driver's torch.distributed.run line
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "3", "--edge-sharded"], 4, {}, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--edge-sharded"]                 # the user's flags reach every rank


def _run(extra):
    env = dict(os.environ, SSLAM_BENCH_LAUNCH_PROBE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                                                       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_by_itself():
    r = _run(["--gpus", "2", "--steps", "3"])
    assert r["n_gpus"] == 2 and r["gpus_arg"] == 2 and r["steps"] == 3 and not r["edge_sharded"]


def test_bench_gpus_2_edge_sharded_flag_reaches_the_ranks():
    r = _run(["--gpus", "2", "--edge-sharded"])
    assert r["n_gpus"] == 2 and r["edge_sharded"]


def test_bench_gpus_1_stays_in_process():
    assert _run(["--gpus", "1"])["n_gpus"] == 1
