cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/gather_mix tools/ubench/gather_mix.hip 2>/dev/null
/tmp/gather_mix > gpurun_out/r06_gather_mix.txt 2>&1
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1      # builds + caches the graph
for rep in 1 2; do
QV_HNSW_DYN=0 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_ab_static_$rep.txt 2>&1
QV_HNSW_DYN=1 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_ab_dyn_$rep.txt 2>&1
done
QV_HNSW_VIS_MULT=128 QV_HNSW_DYN=1 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_ab_dyn_vis128.txt 2>&1
QV_HNSW_WAVES_PER_CU=12 QV_HNSW_DYN=1 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_ab_dyn_w12.txt 2>&1
cat gpurun_out/r06_gather_mix.txt; grep -h nq gpurun_out/r06_ab_*.txt | cut -c1-120
