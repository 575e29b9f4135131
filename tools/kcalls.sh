#!/bin/bash
# Every launch of the kernels whose name contains <substring>, in launch order (start time since the first, duration):
# tools/kcalls.sh <substring> -- <program> <args...>
pat="$1"; shift 2
export TMPDIR=/tmp
d=/tmp/kc_$$; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -o p -- "$@" > /tmp/kc_$$.log 2>&1)
f=$(find $d -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/kc_$$.log; exit 1; }
python3 - "$f" "$pat" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    if sys.argv[2] in r["Kernel_Name"]:
        print("%10.3f ms  %9.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
PY
