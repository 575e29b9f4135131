cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for f in 0 3; do
QV_HNSW_DYN=0 QV_HNSW_FRONT=$f QV_HNSW_WAVES_PER_CU=12 QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_prof.so timeout 600 python3 tools/dev_hnsw_r06.py 6144 128 1 2>&1 | grep "^blk" | tail -2 | cut -c1-400 > gpurun_out/r06_prof_front_$f.txt
done
cat gpurun_out/r06_prof_front_*.txt
