cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for rep in 1 2 3; do for lib in libqv_cas.so libqv.so; do echo "== $lib"; QV_LIB_PATH=$PWD/quiver_amd/lib/$lib python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-100; done; done
