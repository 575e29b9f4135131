cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
rocm-smi --showpower --showclocks --showmaxpower --showtemp > gpurun_out/r06_smi_idle.txt 2>&1
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (edge|junction|memory)" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/r06_smi_hnsw.txt 2>&1 &
python3 tools/dev_hnsw_r06.py 65536,65536,65536,65536 128 3 > gpurun_out/r06_smi_hnsw_run.txt 2>&1
wait
( for i in $(seq 1 16); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (edge|junction|memory)" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/r06_smi_flat.txt 2>&1 &
python3 bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-also > gpurun_out/r06_smi_flat_run.txt 2>&1
wait
head -30 gpurun_out/r06_smi_idle.txt; sed -n 8,14p gpurun_out/r06_smi_hnsw.txt; sed -n 6,10p gpurun_out/r06_smi_flat.txt; grep nq gpurun_out/r06_smi_hnsw_run.txt | cut -c1-100
