cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/ub
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/ub/gather_vmm tools/ubench/gather_vmm.hip 2>/dev/null
echo "== fresh box"; /tmp/ub/gather_vmm
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for i in 1 2 3 4; do python3 tools/dev_hnsw_r06.py 8192 128 3 2>&1 | grep nq | cut -c1-75; done
echo "== after five 10 GB processes"; /tmp/ub/gather_vmm
