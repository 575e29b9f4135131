cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_coverage.py -m gpu -q -k "redo" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
