#!/bin/bash
# k_bf16x1_filter_w8 variants (QV_MFMA_W8_SHAPE: 1 = product, 2 = the whole epilogue after the loop) on 256 x 1M x 768 cosine, product
# library and measurement builds (tools/build_variant.sh)
root=${GRAFT_REPO_ROOT:-$PWD}; tag=${1:-w8}
mkdir -p $root/gpurun_out
for lib in "" $(ls $root/quiver_amd/lib/libqv_*.so 2>/dev/null); do
  name=$(basename "${lib:-libqv_product.so}" .so)
  for shape in ${SHAPES:-1 2}; do
    out=$root/gpurun_out/${tag}_${name}_shape$shape.txt
    QV_MFMA_W8_SHAPE=$shape QV_LIB_PATH=$lib bash $root/tools/ktrace.sh $out -- python3 $root/tools/dev_batched.py cosine 256 ${2:-1000000} 768 10 > /dev/null 2>&1
    echo "== $name shape=$shape"; grep -E "filter_w8" $out | cut -c1-60,73-200 | head -2
  done
done
