#!/bin/bash
# the fp32-MFMA filter (BASELINE configs[2] as written) on 256 x 1M x 768, cosine and dot: batch time + per-kernel times
root=${GRAFT_REPO_ROOT:-$PWD}; tag=${1:-fp32}
mkdir -p $root/gpurun_out
for m in cosine dot_product; do
  DEV_FILTER=fp32 python3 $root/tools/dev_batched.py $m 256 1000000 768 10 2>&1 | grep "batched"
  out=$root/gpurun_out/${tag}_$m.txt
  DEV_FILTER=fp32 bash $root/tools/ktrace.sh $out -- python3 $root/tools/dev_batched.py $m 256 1000000 768 10 > /dev/null 2>&1
  grep -E "filter|rescore|sample|prep|flat_scan" $out | cut -c1-62,73-200 | head -7
done
