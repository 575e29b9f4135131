#!/bin/bash
# Probe of the register-resident-query filter (qv_qreg.hip) on the GPU box: results against the exact scan and per-kernel times
# for float32 rows and the bfloat16 copy, for the product library and any measurement builds present (quiver_amd/lib/libqv_<name>.so,
# tools/build_variant.sh), beside the eight-wave kernels (QV_QREG=2).  bash tools/run_qreg_probe.sh [tag] [metrics...]
tag=${1:-qreg}; shift
metrics=${@:-cosine}
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
for lib in "" $(ls $root/quiver_amd/lib/libqv_*.so 2>/dev/null); do
  name=$(basename "${lib:-libqv_product.so}" .so)
  for bf in 0 1; do
    for qr in 1 2; do
      [ "$qr" = 2 ] && [ -n "$lib" ] && continue
      for m in $metrics; do
        out=$root/gpurun_out/${tag}_${name}_bf${bf}_qreg${qr}_$m.txt
        echo "== $name bf16rows=$bf QV_QREG=$qr $m"
        QV_LIB_PATH=$lib QV_QREG=$qr DEV_BF16_ROWS=$bf DEV_REPS=60 timeout 300 bash $root/tools/ktrace.sh $out -- python3 $root/tools/dev_batched.py $m 256 1000000 768 10 > /tmp/probe.out 2>&1
        grep -E "filter|rescore" $out | cut -c1-60,73-200 | head -3
        grep "batched " /tmp/kt_*.log 2>/dev/null | tail -1 | cut -d: -f2-; rm -f /tmp/kt_*.log
      done
    done
  done
done
