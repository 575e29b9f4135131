#!/bin/bash
# The round's final validation on the GPU box: every -m gpu test, the round's profiles and PMC records, the kernel-coverage map.
#   gpurun --timeout 5400 -- bash tools/run_final.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_gpu_tests.txt
bash tools/run_round_profiles.sh r06 > gpurun_out/r06_round_profiles.log 2>&1
bash tools/run_round_pmc.sh r06 > gpurun_out/r06_round_pmc.log 2>&1
bash tools/run_kernel_coverage.sh $PWD/gpurun_out/r06_kernel_coverage.txt > gpurun_out/r06_cov.log 2>&1
grep -E "^chunk|FAILED|NOT TRACED" gpurun_out/r06_cov.log; head -3 gpurun_out/r06_kernel_coverage.txt
head -c 3000 gpurun_out/r06_bench_n1.jsonl; echo; cat gpurun_out/r06_hnsw_1Mx768.txt; cat gpurun_out/r06_10Mx768_pmc.json
bash tools/run_largek_pmc.sh > gpurun_out/r06_largek_pmc.log 2>&1; tail -3 gpurun_out/r06_largek_pmc.txt
bash tools/run_dispatch_table.sh gpurun_out/r06_dispatch_table.md > /dev/null 2>&1; grep -c "^|" gpurun_out/r06_dispatch_table.md
