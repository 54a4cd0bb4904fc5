#!/bin/bash
# C3 with more of its large halos through the pipeline (SOAP_ENGINE_NB_GIANT: halos above go there)
cd "$(dirname "$0")/.."
rm -f gpurun_out/r2_gc3_*.json
for g in ${LIMITS:-16384 32768 65536 131072 262144}; do
  SOAP_ENGINE_NB_GIANT=$g timeout -k 10 300 python bench.py --workload ${WL:-c3} --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 > gpurun_out/r2_gc3_${g}.json 2> gpurun_out/r2_gc3_${g}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_gc3_*.json'), key=lambda f: int(f.split('_')[-1].split('.')[0])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernel_ms']
        print(f, 'ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f build %.2f A %.2f B %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), d['halos_by_path'])
    except Exception as e:
        print(f,'ERR',e)
PY
