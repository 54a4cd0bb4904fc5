#!/bin/bash
# upper limit of the wavefront class on C3 (SOAP_ENGINE_NB_SMALL; the 256-thread class keeps its 262144)
cd "$(dirname "$0")/.."
for w in ${WLS:-c3}; do
for g in ${LIMITS:-16384 24576 32768 49152}; do
  SOAP_ENGINE_NB_SMALL=$g SOAP_ENGINE_NB_MEDIUM=${MEDIUM:-262144} timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$w NB_SMALL=$g ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024']), d['halos_per_class'], d['single_kernel_passes'])"
done; done
