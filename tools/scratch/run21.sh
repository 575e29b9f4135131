cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
timeout 1800 python3 -m pytest tests/test_gpu_graph.py tests/test_gpu_build.py tests/test_gpu_graph_forms_fuzz.py tests/test_gpu_coverage.py tests/test_gpu_fullsize.py tests/test_gpu_host.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for rep in 1 2 3; do python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-100; done
cp gpurun_out/r06_dispatch_table.md gpurun_out/r06_dispatch_table_prev.md 2>/dev/null
