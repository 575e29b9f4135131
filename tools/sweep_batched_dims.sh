#!/bin/bash
# The default batched path (filter + re-score, 256 queries, k = 10) over ~3 GB of rows at one dimension after the other:
#   tools/sweep_batched_dims.sh [out.txt] [dims...]      ms per batch, GB/s of rows, identical to the exact scan
out=${1:-/dev/stdout}; shift
dims=${@:-64 100 128 192 200 256 300 320 384 768 960 1000 1536}
root=$(cd "$(dirname "$0")/.." && pwd)
for d in $dims; do
  rows=$(( 3000000000 / (4 * d) / 64 * 64 ))
  line=$(DEV_REPS=20 python3 $root/tools/dev_batched.py cosine 256 $rows $d 10 2>/dev/null | head -1)
  ms=$(echo "$line" | sed -n 's/.*: \([0-9.]*\) ms\/batch.*/\1/p')
  same=$(echo "$line" | sed -n 's/.*identical to the exact scan: \(.*\)/\1/p')
  python3 -c "print('dim %5d rows %9d: %7.3f ms/batch  %6.0f GB/s of rows  identical %s' % ($d, $rows, $ms, $rows * $d * 4 / ($ms * 1e-3) / 1e9, '$same'))"
done | tee $out
