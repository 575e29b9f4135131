cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp DEV_REPS=30
out=$root/gpurun_out/r06_largek_after3.txt; : > $out
for k in 100 1000; do
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
for k in 65 128 300 2048 4096; do echo "== k=$k $(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep batched | cut -c1-140)" >> $out; done
for m in dot_product euclidean squared_euclidean; do echo "== $m k=1000 $(python3 tools/dev_batched.py $m 256 1000000 768 1000 2>&1 | grep batched | cut -c1-140)" >> $out; done
echo "== rowmajor k=100/1000 $(DEV_ROWMAJOR=1 python3 tools/dev_batched.py cosine 256 1000000 768 100 2>&1 | grep batched | cut -c1-140) $(DEV_ROWMAJOR=1 python3 tools/dev_batched.py cosine 256 1000000 768 1000 2>&1 | grep batched | cut -c1-140)" >> $out
echo "== 10M k=100 $(python3 tools/dev_batched.py cosine 256 10000000 768 100 2>&1 | grep batched | cut -c1-140)" >> $out
cat $out
timeout 1500 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_batched.py tests/test_gpu_sharded_index.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
