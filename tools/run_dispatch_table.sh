#!/bin/bash
# DESIGN.md 4.1 from traces: every shape of tools/dispatch_table.py in a process of its own under rocprofv3 --kernel-trace.
#   bash tools/run_dispatch_table.sh [out.md]
root=${GRAFT_REPO_ROOT:-$PWD}; out=${1:-$root/gpurun_out/dispatch_table.md}
export TMPDIR=/tmp
d=/tmp/qv_disp_$$; rm -rf $d; mkdir -p $d
for s in $(python3 $root/tools/dispatch_table.py names); do
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d/$s -o t -- python3 $root/tools/dispatch_table.py shape $s > $d/$s.log 2>&1) || echo "shape $s failed: $(tail -1 $d/$s.log)"
done
python3 $root/tools/dispatch_table.py table $d > $out
cat $out; rm -rf $d
