#!/bin/bash
# Repeats of the headline on ONE box (the driver's timed region is 0.45 s: how far do repeats spread?): gpurun -- 'bash tools/bench_repeats.sh [n]'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; N=${1:-5}
cd /tmp && export TMPDIR=/tmp
B="--gpus 1 --warmup 5 --no-cpu-baseline --no-frontend --no-single --plane-batch 0"
{
echo "bench.py $B, one box, $N repeats at --steps 20 (the driver's) and two at --steps 60: value (it/s), ms_per_step, timed seconds"
for k in $(seq 1 $N); do
  timeout 400 python $R/bench.py $B --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20', d['value'], d['ms_per_step'], d['timed_seconds'])"
done
for k in 1 2; do
  timeout 400 python $R/bench.py $B --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 60', d['value'], d['ms_per_step'], d['timed_seconds'], 'iterations per graph', d['iters_min'], d['iters_max'])"
done
} | tee $O/r6_bench_repeats.txt
