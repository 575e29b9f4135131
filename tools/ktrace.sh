#!/bin/bash
# Per-kernel, per-grid-size mean durations of one command:  tools/ktrace.sh <out.txt> -- <program> <args...>
out="$1"; shift 2
export TMPDIR=/tmp
d=/tmp/kt_$$; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -o p -- "$@" > /tmp/kt_$$.log 2>&1)
f=$(find $d -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/kt_$$.log; exit 1; }
python3 - "$f" > "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][:70]
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (name, grid), v in rows[:25]:
    print("%-72s grid %-16s n=%-5d mean %10.1f us  total %10.1f us" % (name, grid, len(v), sum(v) / len(v), sum(v)))
PY
cat "$out"
