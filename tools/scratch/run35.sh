cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_batched.py -x -q -m gpu -k "large_k" 2>&1 | tail -15 | cut -c1-250
