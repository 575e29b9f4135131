cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DEV_REPS=30 QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_dev.so
for k in 10 16 24 32 48 64; do
  a=$(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2 | cut -c1-16)
  b=$(QV_BATCHED_SELECT_FROM=1 python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-120)
  echo "k=$k  wave lists:$a   selection path:$b"
done
