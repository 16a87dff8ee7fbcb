#!/bin/bash
# compaction check (bench headline), tick timing breakdown, backend tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_graph_gpu.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-frontend --plane-batch 0 --no-cpu-baseline > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/bench.json; tail -5 $O/bench_stderr.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
for k in ['value','ms_per_step','trial_rounds','iters_min','iters_max','graphs_terminated','kernel_ms','setup_seconds','single_graph']: print(k, d.get(k))
print({k:v for k,v in d.get('tick_replay',{}).items() if k not in ('workload','cpu_baseline')})
PY
SSLAM_TIMING=1 python $R/tools/tick_timing.py > $O/tick.txt 2> $O/tick_timing.txt; cat $O/tick.txt; grep "optimize:" $O/tick_timing.txt | awk 'NR%10==0' | tail -12; grep "plan build" $O/tick_timing.txt | tail -3
