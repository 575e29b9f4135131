#!/bin/bash
# Large-k batches (256 x 1M x 768, cosine, default layout), before and after round 6: per-kernel times (rocprofv3 --kernel-trace --stats) and the
# memory-side bytes of the exact pass (FETCH_SIZE, TCC_EA0_RDREQ_sum; separate --pmc passes).  "before" = the measurement build with the round-5
# shape switched back on (lane-per-row exact kernel, sample of half the corpus, separate narrowing kernels): QV_LK_*=2 on libqv_dev.so.
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r06_largek_pmc.txt; : > $out
export TMPDIR=/tmp DEV_REPS=20 QV_LIB_PATH=$root/quiver_amd/lib/libqv_dev.so
for k in 100 1000; do for mode in before after; do
  if [ $mode = before ]; then export QV_LK_EXACT_WAVE=2 QV_LK_TILE_PASS=2 QV_LK_GUESS=2 QV_LK_NARROW=2; pat=k_cand_exact; else unset QV_LK_EXACT_WAVE QV_LK_TILE_PASS QV_LK_GUESS QV_LK_NARROW; pat=$([ $k = 100 ] && echo k_cand_exact_wave || echo k_tp_exact); fi
  d=/tmp/lkp_${k}_$mode; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /tmp/lkp.log 2>&1)
  echo "== k=$k $mode: $(grep '^batched' /tmp/lkp.log | cut -c1-130)" >> $out
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
tot = 0.0
rows = list(csv.DictReader(open(sys.argv[1])))
nb = float([r for r in rows if "k_qreg_filter" in r["Name"]][0]["Calls"])    # batches of the process (one main filter launch each)
for r in rows:
    n = r["Name"]
    if any(x in n for x in ("rescore", "cand_", "qreg", "sample", "prep", "select", "filter", "k_tp", "fillBuffer")) and int(r["Calls"]) >= 20:
        per = float(r["TotalDurationNs"]) / nb / 1e3; tot += per
        print("  %-62s launches/batch %4.1f  us/batch %8.1f" % (n[:62], int(r["Calls"]) / nb, per))
print("  sum of the batch's kernels: %.1f us" % tot)
PY
  bash $root/tools/pmc_kernel.sh $pat /tmp/lkp_pmc.txt "FETCH_SIZE/TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /dev/null 2>&1
  sed "s/^/  pmc $pat: /" /tmp/lkp_pmc.txt >> $out
done; done
cat $out
