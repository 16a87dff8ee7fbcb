#!/bin/bash
# Re-measure everything the round's profiles/ hold.  Run on the GPU box: gpurun -- 'bash tools/refresh_profiles.sh [tests]'
# Writes gpurun_out/refresh/r2_*; copy the ones to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "${1:-tests}" = "tests" ]; then
  python -m pytest $R/tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
fi
DRV="--gpus 1 --steps 20 --warmup 5"
python $R/bench.py $DRV > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r2_bench.json; cut -c1-400 $O/r2_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $DRV --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r2_bench_kernel_stats.csv; head -12 $O/r2_bench_kernel_stats.csv | cut -c1-160
FB=$(python -c "import json;print(json.load(open('$O/r2_bench.json'))['roofline_factor']['bytes_per_launch'])")
JB=$(python -c "import json;print(json.load(open('$O/r2_bench.json'))['roofline_jacobian_build']['bytes_per_launch'])")
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_factor_$c -- python $R/tools/prof_opt.py 512 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_jac_$c -- python $R/tools/lin_only.py 512 > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_factor_FETCH_SIZE $O/pmc_factor_WRITE_SIZE $O/r2_pmc_factor.json $FB 0 k_chol_tail k_chol_pieces k_chol_begin k_chol_end
python $R/tools/pmc_traffic.py $O/pmc_jac_FETCH_SIZE $O/pmc_jac_WRITE_SIZE $O/r2_pmc_jacobian_build.json $JB 0 k_linearize_rowthread k_linearize_lm_rows k_linearize_dups
cp $O/r2_pmc_factor.json $O/r2_pmc_jacobian_build.json $R/profiles/   # so that the final bench line quotes this run's traffic
python $R/bench.py $DRV > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r2_bench.json
rm -rf $O/stats $O/pmc_*
