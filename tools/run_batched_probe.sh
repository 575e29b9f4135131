#!/bin/bash
# One probe of the batched path on the GPU box: per-kernel times (rocprofv3 --kernel-trace) of 256 and 64 queries x 1M x 768 on
# float32 rows and on the bfloat16 row copy, for the product library and for any measurement builds present
# (quiver_amd/lib/libqv_<name>.so, tools/build_variant.sh).  bash tools/run_batched_probe.sh [tag]
tag=${1:-probe}
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
for lib in "" $(ls $root/quiver_amd/lib/libqv_*.so 2>/dev/null); do
  name=$(basename "${lib:-libqv_product.so}" .so)
  for cfg in "0 256" "1 256" "1 64" "0 64"; do
    set -- $cfg
    out=$root/gpurun_out/${tag}_${name}_bf$1_q$2.txt
    QV_LIB_PATH=$lib DEV_BF16_ROWS=$1 bash $root/tools/ktrace.sh $out -- python3 $root/tools/dev_batched.py cosine $2 1000000 768 10 > /dev/null 2>&1
    echo "== $name bf16rows=$1 nq=$2"; grep -E "filter|rescore|sample|prep" $out | cut -c1-60,73-200 | head -6
  done
done
