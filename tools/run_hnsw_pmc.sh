#!/bin/bash
# PMC passes over the throughput traversal kernel (k_hnsw_search_wave<0, 4, 4, true, 1>: cosine, efSearch 128, 8192 queries per call,
# 1M x 768 graph): every counter set its own rocprofv3 --kernel-trace --pmc pass (tools/pmc_kernel.sh).
#   bash tools/run_hnsw_pmc.sh <out.txt> [sets]
out=${1:-gpurun_out/hnsw_pmc.txt}; root=${GRAFT_REPO_ROOT:-$PWD}
sets=${2:-"FETCH_SIZE GRBM_GUI_ACTIVE/WRITE_SIZE TCC_HIT_sum TCC_MISS_sum/TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum/SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU/SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM/SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS/TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum/TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum/TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"}
export QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
[ -f $QV_GRAPH_CACHE ] || python3 $root/tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1   # build once, outside the counter passes
PMC_LAST=3 bash $root/tools/pmc_kernel.sh "k_hnsw_search_wave<0, 4, 4, true, 1>" $out "$sets" -- python3 $root/tools/dev_hnsw_r06.py 8192 128 2 > /dev/null 2>&1
cat $out
