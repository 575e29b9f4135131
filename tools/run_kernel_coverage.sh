#!/bin/bash
# Kernel-coverage map: pytest -m gpu under rocprofv3 --kernel-trace (the program directly after `--`), the test files in chunks so
# that one trace stays small; then tools/kernel_coverage.py compares the launched kernel names with the kernels libqv.so ships.
#   bash tools/run_kernel_coverage.sh [out.txt]        (on the GPU box; ~10-15 min)
# Exit status 1 if a shipped kernel was never launched.
root=${GRAFT_REPO_ROOT:-$PWD}; out=${1:-$root/gpurun_out/kernel_coverage.txt}
export TMPDIR=/tmp
cov=/tmp/qv_cov_$$; rm -rf $cov; mkdir -p $cov $(dirname $out)
cd $root
n=0
for chunk in "tests/test_gpu_flat.py tests/test_gpu_small.py tests/test_gpu_scan_split.py tests/test_gpu_select.py" \
             "tests/test_gpu_batched.py tests/test_gpu_mq64.py" \
             "tests/test_gpu_fuzz.py tests/test_gpu_graph_forms_fuzz.py" \
             "tests/test_gpu_graph.py tests/test_gpu_build.py tests/test_gpu_host.py" \
             "tests/test_gpu_concurrent.py tests/test_gpu_sharded.py tests/test_gpu_sharded_abi.py tests/test_gpu_sharded_index.py" \
             "tests/test_gpu_fullsize.py tests/test_derived_kats.py tests/test_gpu_coverage.py"; do
    n=$((n + 1))
    files=""; for f in $chunk; do [ -f $f ] && files="$files $f"; done
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $cov/c$n -o t -- python3 -m pytest -q -m gpu -p no:cacheprovider --rootdir $root $(for f in $files; do echo $root/$f; done) > $cov/c$n.log 2>&1)
    echo "chunk $n rc=$? : $(grep -E "passed|failed" $cov/c$n.log | tail -1)"
    grep -E "^FAILED|^ERROR" $cov/c$n.log | head -20
done
# any -m gpu test file not named above would be missed silently: list them
for f in tests/test_*.py; do grep -q "pytest.mark.gpu" $f && ! grep -q "$(basename $f)" $0 && echo "NOT TRACED: $f"; done
python3 tools/kernel_coverage.py check $cov > $out; rc=$?
head -40 $out
rm -rf $cov
exit $rc
