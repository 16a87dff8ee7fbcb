#!/bin/bash
# full GPU suite + full default bench (driver command line)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 1200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/bench.json; tail -5 $O/bench_stderr.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
for k in ['value','ms_per_step','trial_rounds','iters_min','iters_max','graphs_terminated','kernel_ms','setup_seconds','single_graph','schur_pcg','plane_landmarks','cpu_baseline']: print(k, d.get(k))
print('roofline', d['roofline'])
print('jac', d['roofline_jacobian_build'])
for k in ('tick_replay','tick_replay_long'):
    t=d.get(k,{}); print(k, {a:b for a,b in t.items() if a not in ('workload','cpu_baseline')}, 'CPU:', {a:b for a,b in (t.get('cpu_baseline') or {}).items() if a!='sample'})
PY
