cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_hist.so python3 tools/dev_hnsw_hist.py > gpurun_out/r06_hnsw_hist.txt 2>&1
tail -3 gpurun_out/r06_hnsw_hist.txt
