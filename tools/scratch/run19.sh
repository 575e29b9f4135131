cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
python3 tools/dev_hnsw_state.py 2>&1 | grep trial | tee gpurun_out/r06_hnsw_state.txt
