cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
smi() { rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "junction|memory\)|fclk|mclk|sclk|Power \(W\)|Package Power" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';'; echo; }
date +%s.%N; smi
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for i in 1 2 3 4 5 6 7 8; do date +%s.%N; smi; python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-75; done
echo "sleep 20"; sleep 20
for i in 9 10 11; do date +%s.%N; smi; python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-75; done
