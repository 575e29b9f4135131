cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for lib in libqv.so libqv_cpol2.so libqv_cpol1.so libqv_cpol16.so libqv_cpol3.so libqv.so; do echo "== $lib"; QV_LIB_PATH=$PWD/quiver_amd/lib/$lib python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-100; done
