cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DEV_REPS=30
for k in 10 16 32 64 100 300 1000 4096; do echo "k=$k $(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-120)"; done
for k in 16 64 100 1000; do echo "10M k=$k $(python3 tools/dev_batched.py cosine 256 10000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-120)"; done
timeout 2400 python3 -m pytest tests/test_gpu_batched.py tests/test_gpu_sharded_index.py tests/test_gpu_coverage.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | cut -c1-200
