cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp DEV_REPS=30
for sl in 8 4; do
  export QV_LK_TP_SLAB=$sl
  d=/tmp/lk_$sl; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 1000 > /tmp/lk.log 2>&1)
  echo "== slab $sl $(grep batched /tmp/lk.log | cut -c1-140)"
  f=$(find $d -name "*kernel_stats.csv" | head -1); grep k_tp_exact $f | cut -d, -f1-4 | cut -c1-40,150-
done
unset QV_LK_TP_SLAB
timeout 1200 python3 -m pytest tests/test_gpu_batched.py -x -q -m gpu -k large_k 2>&1 | tail -3 | cut -c1-200
