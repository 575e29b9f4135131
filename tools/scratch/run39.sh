cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DEV_REPS=40 QV_LIB_PATH=$PWD/quiver_amd/lib/libqv_dev.so
for k in 1 5 10 12; do
  a=$(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2 | cut -c1-16)
  b=$(QV_BATCHED_SELECT_FROM=1 python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-110)
  a2=$(python3 tools/dev_batched.py cosine 256 1000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2 | cut -c1-16)
  echo "k=$k  wave lists:$a /$a2   selection path:$b"
done
for k in 10; do
  a=$(python3 tools/dev_batched.py cosine 256 10000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2 | cut -c1-16)
  b=$(QV_BATCHED_SELECT_FROM=1 python3 tools/dev_batched.py cosine 256 10000000 768 $k 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-110)
  echo "10M k=$k  wave lists:$a   selection path:$b"
done
for nq in 16 64; do
  a=$(python3 tools/dev_batched.py cosine $nq 1000000 768 10 2>&1 | grep '^batched' | cut -d: -f2 | cut -c1-16)
  b=$(QV_BATCHED_SELECT_FROM=1 python3 tools/dev_batched.py cosine $nq 1000000 768 10 2>&1 | grep '^batched' | cut -d: -f2- | cut -c1-110)
  echo "nq=$nq k=10  wave lists:$a   selection path:$b"
done
