cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
./tools/ubench/bin/gather_mix > gpurun_out/r06_gather_mix.txt 2>&1
QV_HNSW_DYN=0 python3 tools/dev_hnsw_r06.py 4096,8192,16384,32768,65536 128 3 > gpurun_out/r06_hnsw_nq_static.txt 2>&1
QV_HNSW_DYN=1 python3 tools/dev_hnsw_r06.py 4096,8192,16384,32768,65536 128 3 > gpurun_out/r06_hnsw_nq_dyn.txt 2>&1
QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_spec1.so python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_hnsw_nq_spec1.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_graph.py tests/test_gpu_build.py tests/test_gpu_graph_forms_fuzz.py -m gpu -x -q > gpurun_out/r06_graph_tests.txt 2>&1
tail -2 gpurun_out/r06_graph_tests.txt; cat gpurun_out/r06_gather_mix.txt; tail -5 gpurun_out/r06_hnsw_nq_static.txt gpurun_out/r06_hnsw_nq_dyn.txt gpurun_out/r06_hnsw_nq_spec1.txt
