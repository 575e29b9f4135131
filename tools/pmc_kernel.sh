#!/bin/bash
# Per-kernel PMC means for one command: tools/pmc_kernel.sh <kernel-name-substring> <out.txt> <counter sets separated by '/'> -- <program> <args...>
# Each counter set is its own rocprofv3 pass (--kernel-trace --pmc only).  PMC_LAST=n: only the last n matching dispatches
# (a program whose set-up launches the same kernel instantiation as the measured calls, e.g. a graph build before its searches).
pat="$1"; out="$2"; sets="$3"; shift 4
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
: > "$out"
IFS='/' read -ra SETS <<< "$sets"
n=0
for set in "${SETS[@]}"; do
    n=$((n + 1)); d=/tmp/pmc_$$_$n; rm -rf $d
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $set -d $d -o p -- "$@" > /tmp/pmc_$$_$n.log 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1); [ -z "$f" ] && { tail -5 /tmp/pmc_$$_$n.log; continue; }
    python3 - "$f" "$pat" "${PMC_LAST:-0}" >> "$out" <<'PY'
import csv, sys, collections
f, pat, last = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if last: v = v[-last:]
    print("%-34s mean %.6g over %d launches" % (k, sum(v) / len(v), len(v)))
PY
done
cat "$out"
