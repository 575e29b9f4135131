cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp DEV_REPS=30
timeout 1200 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_batched.py -x -q -m gpu 2>&1 | tail -5
out=$root/gpurun_out/r06_largek_after2.txt; : > $out
for k in 300 1000 4096; do
  d=/tmp/lk_${k}; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /tmp/lk.log 2>&1)
  echo "== k=$k tiles  $(grep batched /tmp/lk.log | cut -c1-140)" >> $out
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(x in n for x in ("rescore", "cand_", "qreg", "sample", "prep", "select", "filter", "k_tp", "fillBuffer")) and int(r["Calls"]) >= 30:
        print("  %-70s calls %4s avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
for k in 300 1000; do echo "== k=$k gather (QV_LK_TILE_PASS=2)" >> $out; QV_LIB=$root/quiver_amd/lib/libqv_dev.so QV_LK_TILE_PASS=2 python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep batched >> $out; 
 echo "== k=$k pass (QV_LK_TILE_PASS=3)" >> $out; QV_LIB=$root/quiver_amd/lib/libqv_dev.so QV_LK_TILE_PASS=3 python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep batched >> $out; done
cat $out
