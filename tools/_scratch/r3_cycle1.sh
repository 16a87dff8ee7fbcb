#!/bin/bash
# Round-3 first GPU round trip: full GPU parity suite, the bench with 512 distinct graphs (baseline before the new factorisation), and
# the MFMA / FP64 instruction-mix counters of the LM kernels (north_star: "MFMA only if ... evidenced by rocprof MFMA utilisation").
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/r3_base_bench.json; cut -c1-600 $O/r3_base_bench.json; tail -3 $O/bench_stderr.txt
rm -f $O/r3_pmc_mfma.txt
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32"; do
  rm -rf $O/_pmc
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/_pmc -- python $R/tools/prof_opt.py 64 1 > $O/_pmc.log 2>&1 || { echo "set failed: $set"; tail -3 $O/_pmc.log; }
  python $R/tools/pmc_summary.py $O/_pmc 2>/dev/null | grep -E "k_chol|linearize|k_chi2" >> $O/r3_pmc_mfma.txt
  rm -rf $O/_pmc
done
sort $O/r3_pmc_mfma.txt | cut -c1-300
