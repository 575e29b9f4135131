cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1      # builds + caches the graph
rm -f gpurun_out/r06_front_*
for rep in 1 2; do
for f in 4 1 3; do for w in 12 16; do
QV_HNSW_FRONT=$f QV_HNSW_WAVES_PER_CU=$w python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_front_${f}_w${w}_$rep.txt 2>&1
done; done; done
QV_HNSW_FRONT=4 QV_HNSW_DYN=2 QV_HNSW_WAVES_PER_CU=16 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_front_4_w16_static.txt 2>&1
QV_HNSW_FRONT=3 QV_HNSW_DYN=2 QV_HNSW_WAVES_PER_CU=12 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_front_3_w12_static.txt 2>&1
QV_HNSW_FRONT=3 QV_HNSW_WAVES_PER_CU=10 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_front_3_w10_1.txt 2>&1
QV_HNSW_FRONT=3 QV_HNSW_WAVES_PER_CU=13 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_front_3_w13_1.txt 2>&1
grep -H nq gpurun_out/r06_front_*.txt | cut -c1-120
