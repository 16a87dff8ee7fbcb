#!/bin/bash
# Re-measure everything the round's profiles/ hold.  Run on the GPU box: gpurun -- 'bash tools/refresh_profiles.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -m pytest $R/tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r1_bench.json; cut -c1-400 $O/r1_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r1_bench_kernel_stats.csv; head -8 $O/r1_bench_kernel_stats.csv | cut -c1-160
FB=$(python -c "import json;print(json.load(open('$O/r1_bench.json'))['roofline']['bytes_per_launch'])")
JB=$(python -c "import json;print(json.load(open('$O/r1_bench.json'))['roofline_jacobian_build']['bytes_per_launch'])")
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_factor_$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-frontend --no-single > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_jac_$c -- python $R/tools/lin_only.py 512 > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_factor_FETCH_SIZE $O/pmc_factor_WRITE_SIZE $O/r1_pmc_factor.json $FB 1 k_chol_level k_chol_tail k_chol_begin k_chol_end
python $R/tools/pmc_traffic.py $O/pmc_jac_FETCH_SIZE $O/pmc_jac_WRITE_SIZE $O/r1_pmc_jacobian_build.json $JB 0 k_linearize_rowthread k_linearize_lm_rows k_linearize_dups
cp $O/r1_pmc_factor.json $O/r1_pmc_jacobian_build.json $R/profiles/   # so that the final bench line quotes this run's traffic
python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r1_bench.json
rm -rf $O/stats $O/pmc_*
