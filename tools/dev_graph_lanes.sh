#!/bin/bash
# caller-count sweep on the 1M x 768 graph by traversal lanes (QV_GRAPH_LANES) and largest shared batch (QV_GRAPH_MAX_GROUP)
root=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "1 4096" "2 4096" "4 4096" "1 256" "2 256" "4 256" "2 128" "4 128" "8 128"; do set -- $cfg
  echo -n "lanes $1 max group $2: "
  QV_GRAPH_LANES=$1 QV_GRAPH_MAX_GROUP=$2 python3 $root/tools/bench_callers.py --rows 0 --graph-rows 1000000 --graph-callers 1,8,64,256,1024 --seconds 1.2 | grep "^graph {" | python3 -c "
import sys, json
print('  '.join('%d: %.1fk p50 %.2f p99 %.1f ms' % (e['callers'], e['qps'] / 1e3, e['p50_us'] / 1e3, e['p99_us'] / 1e3) for e in (json.loads(l.split(' ', 1)[1].strip()) for l in sys.stdin)))"
done
