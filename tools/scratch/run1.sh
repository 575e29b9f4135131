cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
python3 tools/dev_hnsw_r06.py 4096,8192,16384,32768,65536 128 3 > gpurun_out/r06_hnsw_nq_sweep.txt 2>&1
QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_prof.so timeout 600 python3 tools/dev_hnsw_r06.py 8192 128 1 > gpurun_out/r06_hnsw_prof_full.txt 2>&1
timeout 1800 bash tools/run_hnsw_pmc.sh gpurun_out/r06_hnsw_pmc_raw.txt > gpurun_out/r06_hnsw_pmc.log 2>&1
tail -3 gpurun_out/r06_hnsw_nq_sweep.txt
