#!/bin/bash
# caller-count sweep on small and medium flat collections, by lane count (QV_FLAT_LANES) and with the sharing off (QV_COALESCE=0)
root=${GRAFT_REPO_ROOT:-$PWD}
for shape in "10000 128" "12000 768" "30000 768" "100000 768"; do set -- $shape
  for L in off 1 2 4 8 auto; do
    if [ $L = off ]; then export QV_COALESCE=0; unset QV_FLAT_LANES; elif [ $L = auto ]; then unset QV_COALESCE; unset QV_FLAT_LANES; else unset QV_COALESCE; export QV_FLAT_LANES=$L; fi
    echo -n "rows $1 dim $2 lanes $L: "
    python3 $root/tools/bench_callers.py --rows $1 --dim $2 --graph-rows 0 --flat-callers 1,8,64,256 --seconds 0.7 | grep callers | python3 -c "
import sys, json
print('  '.join('%d: %.0fk p50 %.0f us' % (e['callers'], e['qps'] / 1e3, e['p50_us']) for e in (json.loads(l.split(' ', 1)[1].strip()) for l in sys.stdin)))"
  done
done
