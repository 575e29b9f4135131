cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
for rep in 1 2; do for lib in libqv.so libqv_ilv.so; do echo "== $lib"; QV_LIB_PATH=$PWD/quiver_amd/lib/$lib bash tools/run_fp32_probe.sh fp32_$lib 2>&1 | grep -E "batched|k_mfma_filter"; done; done > gpurun_out/r06_fp32_ilv.txt 2>&1
cat gpurun_out/r06_fp32_ilv.txt
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
for rep in 1 2; do for lib in libqv.so libqv_nocas.so; do echo "== $lib"; QV_LIB_PATH=$PWD/quiver_amd/lib/$lib python3 tools/dev_hnsw_r06.py 8192,32768 128 3 2>&1 | grep nq | cut -c1-100; done; done > gpurun_out/r06_hnsw_nocas.txt 2>&1
cat gpurun_out/r06_hnsw_nocas.txt
bash tools/run_dispatch_table.sh $PWD/gpurun_out/r06_dispatch_table.md > gpurun_out/r06_dispatch.log 2>&1
grep -c "^|" gpurun_out/r06_dispatch_table.md; grep "failed" gpurun_out/r06_dispatch.log | head
