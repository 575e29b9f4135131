root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
for lib in "" $root/quiver_amd/lib/libqv_epi1.so $root/quiver_amd/lib/libqv_epi3.so; do
  name=$(basename "${lib:-libqv_product.so}" .so)
  QV_LIB_PATH=$lib bash $root/tools/pmc_kernel.sh k_bf16x3_filter_shared $root/gpurun_out/r03_pmc_$name.txt "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES/SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM/SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1
  echo "== $name"; cat $root/gpurun_out/r03_pmc_$name.txt
done
