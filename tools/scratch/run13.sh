cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_coverage.py "tests/test_gpu_graph.py::test_wave_per_query_form_with_the_round6_front_identical_to_oracle" -m gpu -q --durations=5 > gpurun_out/r06_cov_tests.txt 2>&1
grep -E "passed|failed|^FAILED|^E  " gpurun_out/r06_cov_tests.txt | head -30
bash tools/run_kernel_coverage.sh $PWD/gpurun_out/r06_kernel_coverage.txt > gpurun_out/r06_cov.log 2>&1
grep -E "^chunk|FAILED|NOT TRACED" gpurun_out/r06_cov.log; head -3 gpurun_out/r06_kernel_coverage.txt; grep UNLAUNCHED gpurun_out/r06_kernel_coverage.txt
