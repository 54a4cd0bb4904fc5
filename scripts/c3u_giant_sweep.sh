#!/bin/bash
# the uncapped C3 box (halos up to 1e6 particles): which halos go through the multi-workgroup pipeline
for ng in default 200000 131072 100000 65536; do
  if [ $ng = default ]; then unset SOAP_ENGINE_NB_GIANT; else export SOAP_ENGINE_NB_GIANT=$ng; fi
  python bench.py --workload c3u --no-cpu-baseline --no-full-set --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']
print('NB_GIANT=$ng', 'ms/pass %.1f halo %.1f wg256 %.1f wg1024 %.1f wave %.1f build %.1f a %.1f b %.1f'%(j['ms_per_pass'],k['halo'],k['halo_wg256'],k['halo_wg1024'],k['halo_wave'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), j['halos_per_class'], j['halos_by_path'])"
done
