#!/bin/bash
# counters of one filter kernel on 256 x 1M x 768: bash tools/run_pmc_probe.sh <kernel-substring> [lib names...]
root=${GRAFT_REPO_ROOT:-$PWD}; pat=${1:-filter_w8}; shift
mkdir -p $root/gpurun_out
for name in product "$@"; do
  lib=""; [ "$name" != product ] && lib=$root/quiver_amd/lib/libqv_$name.so
  QV_MFMA_W8_SHAPE=${W8_SHAPE:-1} QV_LIB_PATH=$lib bash $root/tools/pmc_kernel.sh $pat $root/gpurun_out/r03_pmc_$name.txt "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM/SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT/GRBM_GUI_ACTIVE GRBM_COUNT/SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1
  echo "== $name"; cat $root/gpurun_out/r03_pmc_$name.txt
done
