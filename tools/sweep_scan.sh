for wg in ${WGS:-1 2}; do for u in ${US:-16 24 32}; do
  echo "wg=$wg u=$u"; QV_SCAN_WG_PER_CU=$wg QV_SCAN_UNROLL=$u python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  10M: qps %.1f kern %.3f ms frac %.3f | 1M: qps %.0f kern %.4f ms frac %.3f'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['also']['qps'],d['also']['scan_kernel_ms'],d['also']['hbm_frac']))
"; done; done
