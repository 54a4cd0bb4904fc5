#!/bin/bash
# C3 under the pipeline modes and a few splits between the persistent wave groups and the pipeline
cd "$(dirname "$0")/.."
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ${WL:-c3} --no-cpu-baseline --steps 2 --warmup 1 --passes-per-step 4 > gpurun_out/r2_m_${tag}.json 2> gpurun_out/r2_m_${tag}.err
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/r2_m_{tag}.json').read().strip().splitlines()[-1]); k=d['kernel_ms']
    print(tag, 'ms_per_pass %.2f halo %.2f wave %.2f wg256 %.2f wg1024 %.2f build %.2f A %.2f B %.2f'%(d['ms_per_pass'],k['halo'],k['halo_wave'],k['halo_wg256'],k['halo_wg1024'],k['pipe_build'],k['pipe_stage_a'],k['pipe_stage_b']), d['halos_by_path'], flush=True)
except Exception as e:
    print(tag,'ERR',e, flush=True)
PY
}
run default A=1
run mode1 SOAP_ENGINE_PIPELINE=1
run mode2 SOAP_ENGINE_PIPELINE=2
run g16k_G0_1024 SOAP_ENGINE_NB_GIANT=16384 SOAP_ENGINE_G0=1024
run g16k_G0_1536 SOAP_ENGINE_NB_GIANT=16384 SOAP_ENGINE_G0=1536
run g64k_G0_1536 SOAP_ENGINE_NB_GIANT=65536 SOAP_ENGINE_G0=1536
