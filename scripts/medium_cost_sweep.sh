#!/bin/bash
# upper limit of the 256-thread class against the pipeline: slowest rank of the C3 box and the small workloads
cd "$(dirname "$0")/.."
for f in ${COSTS:-6e-8 1e-7 1.5e-7 2.5e-7}; do
  for n in ${NS:-2 4 8}; do
    echo "cost $f: $(SOAP_ENGINE_MEDIUM_COST=$f timeout -k 10 400 python scripts/rank_share.py $n c3 2>&1 | grep '^N=')"
  done
  for w in c2 giant; do
    SOAP_ENGINE_MEDIUM_COST=$f timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('cost $f: $w ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f build %.2f A %.2f B %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), d['halos_by_path'])"
  done
done
