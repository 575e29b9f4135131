cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
bash tools/run_kernel_coverage.sh $PWD/gpurun_out/r06_kernel_coverage_first.txt > gpurun_out/r06_cov.log 2>&1
tail -5 gpurun_out/r06_cov.log
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
PMC_LAST=3 bash tools/pmc_kernel.sh "k_hnsw_search_wave<0, 4, 4, true, 1>" $PWD/gpurun_out/r06_hnsw_front_pmc_raw.txt "FETCH_SIZE GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY/WRITE_SIZE TCC_HIT_sum TCC_MISS_sum/TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_EA0_RDREQ_LEVEL_sum/SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU/SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS/TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum/TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" -- python3 $PWD/tools/dev_hnsw_r06.py 8192 128 2 > gpurun_out/r06_hnsw_front_pmc.log 2>&1
cat gpurun_out/r06_hnsw_front_pmc_raw.txt
