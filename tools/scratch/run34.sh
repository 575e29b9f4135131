cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_batched.py tests/test_gpu_sharded_index.py tests/test_gpu_coverage.py tests/test_gpu_sharded.py tests/test_gpu_sharded_abi.py tests/test_gpu_flat.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | cut -c1-200
