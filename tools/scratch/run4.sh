cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
mkdir -p /tmp/ub && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/ub/gather_mix tools/ubench/gather_mix.hip > gpurun_out/r06_gather_mix_build.log 2>&1
ls -la /tmp/ub >> gpurun_out/r06_gather_mix_build.log 2>&1
/tmp/ub/gather_mix > gpurun_out/r06_gather_mix.txt 2>&1
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1      # builds + caches the graph
for w in 16 14 12 10 8; do
QV_HNSW_WAVES_PER_CU=$w python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_waves_$w.txt 2>&1
done
QV_HNSW_VIS_MULT=128 QV_HNSW_WAVES_PER_CU=12 python3 tools/dev_hnsw_r06.py 8192,32768 128 3 > gpurun_out/r06_waves_12_vis128.txt 2>&1
tail -5 gpurun_out/r06_gather_mix_build.log; cat gpurun_out/r06_gather_mix.txt; grep -H nq gpurun_out/r06_waves_*.txt | cut -c1-150
