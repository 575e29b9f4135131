#!/bin/bash
# Which counter, if any, follows the traversal's per-process state (profiles/LAB_r06.md §1 "States")?  Identical processes, each
# under rocprofv3 --kernel-trace --pmc <instruction-cache counters>: the last three k_hnsw_search_wave dispatches' duration beside their counters.
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r06_hnsw_state_pmc.txt; : > $out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 $root/tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1          # builds the graph once, cached
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC_[A-Z_0-9]*ICACHE[A-Z_0-9]*|SQ_IFETCH[A-Z_0-9]*|SQC_INST[A-Z_0-9]*|SQ_INST_LEVEL[A-Z_0-9]*|SQ_WAIT_INST_ANY|SQ_IFETCH" | sort -u | tr '\n' ' ' >> $out; echo >> $out
SETS="${SETS:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES}"
for i in 1 2 3 4 5 6 7 8; do
  d=/tmp/ic_$i; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $SETS -d $d -o p -- python3 $root/tools/dev_hnsw_r06.py 8192 128 3 > /tmp/ic_$i.log 2>&1)
  python3 - "$d" "$i" >> $out <<'PY'
import csv, sys, glob, collections
d, i = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc: print("process %s: no counter file" % i); sys.exit(0)
rows = [r for r in csv.DictReader(open(cc[0])) if "k_hnsw_search_wave" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-3:]
acc = collections.defaultdict(float)
for r in rows:
    if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids)
dur = []
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "k_hnsw_search_wave" in r["Kernel_Name"]: dur.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
dur = [x for _, x in sorted(dur)[-3:]]
print("process %s: kernel ms %s  %s" % (i, " ".join("%.2f" % x for x in dur), "  ".join("%s %.4g" % kv for kv in sorted(acc.items()))))
PY
done
cat $out
