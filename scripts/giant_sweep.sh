#!/bin/bash
# giants mode (SOAP_ENGINE_PIPELINE=3): which halos go through the multi-workgroup build
cd "$(dirname "$0")/.."
for g in 65536 131072 262144; do
  for w in c3 c2 giant; do
    SOAP_ENGINE_NB_GIANT=$g timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 > gpurun_out/r2_g${g}_${w}.json 2> gpurun_out/r2_g${g}_${w}.err
  done
done
SOAP_ENGINE_PIPELINE=0 timeout -k 10 300 python bench.py --workload giant --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 2 > gpurun_out/r2_g0_giant.json 2> gpurun_out/r2_g0_giant.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_g*_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernel_ms']
        print(f, 'ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f build %.2f A %.2f B %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), d['halos_by_path'])
    except Exception as e:
        print(f,'ERR',e)
PY
