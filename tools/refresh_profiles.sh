#!/bin/bash
# Re-measure what the round's profiles/ hold.  Run on the GPU box: gpurun -- 'bash tools/refresh_profiles.sh [sections]'
# sections (default "tests bench stats pmc frontend trace"): tests bench stats pmc frontend trace tick smoke
# Writes gpurun_out/refresh/r6_*; copy the ones to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SECTIONS="${*:-tests bench stats pmc frontend trace}"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
DRV="--gpus 1 --steps 20 --warmup 5"
NB=${NB:-1024}   # graphs of the bench batch (bench.py --batch default): the PMC passes and the launch trace run the same batch
if has tests; then
  python -m pytest $R/tests -m gpu -q > $O/r6_pytest_gpu.txt 2>&1; tail -3 $O/r6_pytest_gpu.txt
fi
if has bench; then
  python $R/bench.py $DRV > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r6_bench.json; cut -c1-400 $O/r6_bench.json
fi
if has stats; then
  # kernel stats of the pass the rooflines come from: one batch-synchronous batch on one stream (--streams 1); in the stream-group
  # region the launches of four parts overlap on the chip and a per-launch duration says nothing about a kernel
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $DRV --streams 1 --no-cpu-baseline --no-frontend --no-single --plane-batch 0 > /dev/null 2>&1
  cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r6_bench_kernel_stats.csv; head -12 $O/r6_bench_kernel_stats.csv | cut -c1-160
  rm -rf $O/stats
fi
if has pmc; then
  BJ=$O/r6_bench.json; [ -f $BJ ] || BJ=$R/profiles/r6_bench.json     # (a run without the bench section: the committed line)
  FB=$(python -c "import json;print(json.load(open('$BJ'))['roofline_factor']['bytes_per_launch'])")
  JB=$(python -c "import json;print(json.load(open('$BJ'))['roofline_jacobian_build']['bytes_per_launch'])")
  # PMC traffic: one full-batch factorisation / Jacobian build of the SAME $NB distinct graphs (tools/pmc_workload.py), separate passes
  # PMC_WHAT="factor" or "jacobian" restricts the passes to one of the two
  for w in ${PMC_WHAT:-factor jacobian}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python $R/tools/pmc_workload.py $NB $w > /dev/null 2>&1
    done
    if [ $w = factor ]; then
      python $R/tools/pmc_traffic.py $O/pmc_factor_FETCH_SIZE $O/pmc_factor_WRITE_SIZE $O/r6_pmc_factor.json $FB 0 k_front_tail k_front_pieces k_chol_tail k_chol_pieces k_chol_begin k_chol_end
    else
      python $R/tools/pmc_traffic.py $O/pmc_jacobian_FETCH_SIZE $O/pmc_jacobian_WRITE_SIZE $O/r6_pmc_jacobian_build.json $JB 0 k_linearize_rowthread k_linearize_lm_rows k_linearize_dups
    fi
  done
  rm -rf $O/pmc_*
fi
if has frontend; then
  # per-kernel times of one batched frontend call (32 frames x 32 boxes)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats -- python $R/tools/frontend_kernels.py > /dev/null 2>&1
  cp $(ls $O/fstats/*/*kernel_stats.csv | head -1) $O/r6_frontend_kernel_stats.csv; head -14 $O/r6_frontend_kernel_stats.csv | cut -c1-140
  rm -rf $O/fstats
fi
if has trace; then
  # per-launch trace of one LM step of the batch (which launches the factorisation is made of and what each takes)
  rm -rf $O/_ptr
  rocprofv3 --kernel-trace --output-format csv -d $O/_ptr -- python $R/tools/prof_opt.py $NB 2 > /dev/null 2>&1
  python $R/tools/level_profile.py $O/_ptr > $O/r6_factor_launch_trace_$NB.txt; rm -rf $O/_ptr
  head -20 $O/r6_factor_launch_trace_$NB.txt | cut -c1-120
fi
if has tick; then
  # kernel stats of the 110-keyframe tick replay (no CPU baseline inside the profiled process)
  cat > /tmp/tick110.py <<PY
import sys, json
sys.path.insert(0, "$R")
import bench
t = bench.bench_tick(0, cpu_baseline=False)
print(json.dumps({k: t[k] for k in ("keyframes", "ms_per_tick", "ms_per_tick_optimize", "ms_per_tick_marginals", "lm_iterations_per_tick")}))
PY
  rm -rf $O/_tk
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/_tk -- python /tmp/tick110.py > $O/tick_stdout.txt 2>&1
  cp $(ls $O/_tk/*/*kernel_stats.csv | head -1) $O/r6_tick_kernel_stats.csv; rm -rf $O/_tk
  tail -1 $O/tick_stdout.txt | cut -c1-200; head -6 $O/r6_tick_kernel_stats.csv | cut -c1-150
fi
if has smoke; then
  (cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") 2>&1 | tail -2
fi
