cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_coverage.py "tests/test_gpu_graph.py::test_wave_per_query_form_with_the_round6_front_identical_to_oracle" -m gpu -q --durations=8 > gpurun_out/r06_cov_tests.txt 2>&1
tail -25 gpurun_out/r06_cov_tests.txt
timeout 900 python3 -m pytest tests/test_gpu_batched.py -m gpu -q -k "bf16_row_plane or query_resident" > gpurun_out/r06_cov_tests2.txt 2>&1
tail -5 gpurun_out/r06_cov_tests2.txt
for lib in libqv.so libqv_wps2.so; do echo "== $lib"; QV_LIB_PATH=$PWD/quiver_amd/lib/$lib bash tools/run_fp32_probe.sh fp32_$lib 2>&1 | tail -12; done > gpurun_out/r06_fp32_wps.txt 2>&1
cat gpurun_out/r06_fp32_wps.txt
