#!/bin/bash
# rank shares of the C3 box at N = 2, 4, 8 (slowest rank decides) and the one-GPU workloads, for one setting of the
# giants' criterion (SOAP_ENGINE_GIANT_FACTOR)
cd "$(dirname "$0")/.."
for n in 2 4 8; do timeout -k 10 400 python scripts/rank_share.py $n c3 2>&1 | grep "^N=\|^rank" | cut -c1-120; done
for w in c3 c2 c5 giant; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$w ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f build %.2f A %.2f B %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), d['halos_by_path'])"
done
