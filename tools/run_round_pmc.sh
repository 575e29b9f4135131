#!/bin/bash
# Round PMC passes (each counter set its own rocprofv3 --kernel-trace --pmc run, tools/pmc_kernel.sh):
#   headline k_flat_scan: FETCH_SIZE / WRITE_SIZE;   default batched filter (k_qreg_filter on float32 rows and on the bfloat16 copy): L1<->L2 requests, L2 hits, HBM fetch, matrix-pipe busy, clocks
# bash tools/run_round_pmc.sh r03
tag=${1:-rXX}; root=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $root/gpurun_out
bash $root/tools/pmc_kernel.sh k_flat_scan $root/gpurun_out/${tag}_pmc_flat_scan.txt "FETCH_SIZE/WRITE_SIZE" -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also > /dev/null 2>&1
bash $root/tools/pmc_kernel.sh qreg_filter $root/gpurun_out/${tag}_pmc_filter_w8.txt "FETCH_SIZE/WRITE_SIZE/TCP_TCC_READ_REQ_sum/TCC_HIT_sum TCC_MISS_sum/TCC_REQ_sum/SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU/GRBM_GUI_ACTIVE/TCP_PENDING_STALL_CYCLES_sum/TCP_GATE_EN1_sum" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1
DEV_BF16_ROWS=1 bash $root/tools/pmc_kernel.sh qreg_filter $root/gpurun_out/${tag}_pmc_bf16rows.txt "FETCH_SIZE/TCP_TCC_READ_REQ_sum/TCC_HIT_sum TCC_MISS_sum/SQ_VALU_MFMA_BUSY_CYCLES/GRBM_GUI_ACTIVE" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1
python3 $root/tools/make_pmc_json.py $root/gpurun_out/${tag}_pmc_flat_scan.txt $tag > $root/gpurun_out/${tag}_10Mx768_pmc.json 2>/dev/null
for f in flat_scan filter_w8 bf16rows; do echo "== $f"; cat $root/gpurun_out/${tag}_pmc_$f.txt; done
