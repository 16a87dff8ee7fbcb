#!/bin/bash
# Round-3 GPU round trip for the window Cholesky: backend parity tests, then bench (short) with per-kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_graph_gpu.py $R/tests/test_slam_gpu.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-frontend --plane-batch 0 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/bench.json; tail -5 $O/bench_stderr.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
for k in ['value','ms_per_step','trial_rounds','iters_min','iters_max','graphs_terminated','kernel_ms','setup_seconds','single_graph','chi2_after']: print(k, d.get(k))
print(d.get('tick_replay'))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/prof_opt.py 512 2 > $O/prof.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; head -14 $O/kernel_stats.csv | cut -c1-200; rm -rf $O/stats
