#!/bin/bash
# the one-term filter with per-row / per-query margins: kernel times at several sample sizes, f32 rows and bf16 plane, 256 and 64 queries
tag=${1:-margin}; root=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $root/gpurun_out
for sr in ${SAMPLES:-0 16384 8192}; do
  for cfg in "0 256 cosine" "1 256 cosine" "1 64 cosine" "0 256 dot_product" "0 256 euclidean"; do
    set -- $cfg
    out=$root/gpurun_out/${tag}_s${sr}_bf$1_q$2_$3.txt
    QV_MFMA_SAMPLE_ROWS=$sr DEV_BF16_ROWS=$1 bash $root/tools/ktrace.sh $out -- python3 $root/tools/dev_batched.py $3 $2 1000000 768 10 > $out.log 2>&1
    echo "== sample=$sr bf16rows=$1 nq=$2 $3"; grep -E "ms/batch" $out.log | head -1; grep -E "filter|rescore|sample|prep" $out | cut -c1-60,73-200 | head -6
  done
done
