#!/bin/bash
# k_rescore_select / k_cand_exact: kernel time (rocprofv3 --kernel-trace --stats) and HBM / L2 request counters, lane-per-row rounds (QV_MFMA_RESCORE_GATHER=2)
# against cooperative group gathers (default), at k = 10 and k = 100
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r05_rescore_probe.txt; : > $out
export TMPDIR=/tmp DEV_REPS=30
for k in 10 100; do for g in 2 1; do
  export QV_MFMA_RESCORE_GATHER=$g
  d=/tmp/rsp_${k}_$g; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /tmp/rsp.log 2>&1)
  echo "== k=$k gather=$g  $(grep batched /tmp/rsp.log | cut -c1-60)" >> $out
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(x in n for x in ("rescore", "cand_exact", "qreg", "sample", "prep", "cand_", "select")):
        print("  %-60s calls %4s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  pat=$([ $k = 10 ] && echo k_rescore_select || echo k_cand_exact)
  bash $root/tools/pmc_kernel.sh $pat /tmp/rsp_pmc.txt "FETCH_SIZE/TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum/TCP_TCC_READ_REQ_sum" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 $k > /dev/null 2>&1
  sed 's/^/  pmc /' /tmp/rsp_pmc.txt >> $out
done; done
cat $out
