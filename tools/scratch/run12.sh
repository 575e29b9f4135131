cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
T="tests/test_gpu_graph.py::test_wave_per_query_form_with_the_round6_front_identical_to_oracle"
echo "== product"; timeout 900 python3 -m pytest "$T" -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
echo "== dev lib, front off"; QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_dev.so QV_HNSW_FRONT=4 timeout 900 python3 -m pytest "$T" -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
echo "== dev lib, front 1 (no spec)"; QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_dev.so QV_HNSW_FRONT=1 timeout 900 python3 -m pytest "$T" -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
echo "== product, one by one"
for kk in "dot-160" "l1-32" "cosine-64-16-64-3-2" "l2sq-128"; do timeout 300 python3 -m pytest "$T" -m gpu -q -k "$kk" 2>&1 | grep -E "passed|failed|^E  " | head -4; done
