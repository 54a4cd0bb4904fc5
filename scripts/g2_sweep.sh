#!/bin/bash
# persistent groups of the 1024-thread class (the slab pre-sizing halves them while it doubles the slab)
cd "$(dirname "$0")/.."
for w in ${WLS:-c5 c3}; do
for g in ${GS:-128 256 512}; do
  SOAP_ENGINE_G2=$g timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$w G2=$g ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024']), d['halos_per_class'])"
done; done
