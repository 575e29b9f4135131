cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
QV_TRACE=1 python3 -c "
import sys; sys.path.insert(0,'.')
from tests.bench import bench_hnsw_build as B
r = B.run(rows=1_000_000, max_level=1, efs=(64,128), cpu_queries=0, intrinsic_dim=16)
for e in r['search']:
    t = e['graph_traversal']; print('ef', e['ef_search'], 'qps', round(t['qps_device_resident']), 'GBps', round(t['gathered_GBps']), 'evals', round(t['evals_per_query']), 'recall', e['search_complete']['recall_at_10_vs_exact'])
" 2>&1 | grep -E "hubs|^ef" | head -20
