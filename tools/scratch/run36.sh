cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DEV_REPS=30; root=$PWD
timeout 2400 python3 -m pytest tests/test_gpu_batched.py -x -q -m gpu -k "large_k" 2>&1 | tail -4 | cut -c1-250
for k in 100 1000; do
  d=/tmp/lk_${k}; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /tmp/lk.log 2>&1)
  echo "== k=$k tiles  $(grep batched /tmp/lk.log | cut -c1-140)"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(x in n for x in ("rescore", "cand_", "qreg", "sample", "prep", "select", "filter", "k_tp", "fillBuffer")) and int(r["Calls"]) >= 30:
        print("  %-70s calls %4s avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
for m in dot_product euclidean squared_euclidean; do echo "== $m k=300 $(python3 tools/dev_batched.py $m 256 1000000 768 300 2>&1 | grep batched | cut -c1-140)"; done
