cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; root=$PWD
export TMPDIR=/tmp DEV_REPS=20
for k in 100 1000; do echo "== 10M k=$k $(python3 tools/dev_batched.py cosine 256 10000000 768 $k 2>&1 | grep batched | cut -c1-140)"; done
for k in 100 1000; do echo "== 1M k=$k $(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep batched | cut -c1-140)"; done
timeout 1500 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_batched.py tests/test_gpu_sharded_index.py tests/test_gpu_coverage.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | cut -c1-200
