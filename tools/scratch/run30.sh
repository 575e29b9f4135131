cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp DEV_REPS=30
timeout 1200 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_batched.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-200
for lib in libqv.so libqv_qbc.so; do
  d=/tmp/lk_$lib; rm -rf $d
  export QV_LIB_PATH=$root/quiver_amd/lib/$lib
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 1000 > /tmp/lk.log 2>&1)
  echo "== $lib $(grep batched /tmp/lk.log | cut -c1-140)"
  f=$(find $d -name "*kernel_stats.csv" | head -1); grep k_tp_exact $f | cut -d, -f2-4
done
