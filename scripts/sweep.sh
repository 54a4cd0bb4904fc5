#!/bin/bash
# threshold sweep of the class boundaries (estimated sphere content) on the C3 workload
# usage: CFGS="8192:131072 12288:131072" bash scripts/sweep.sh
for cfg in ${CFGS:-8192:131072 6144:131072 4096:131072 12288:131072 16384:131072}; do
  a=${cfg%%:*}; b=${cfg##*:}
  SOAP_ENGINE_NB_SMALL=$a SOAP_ENGINE_NB_MEDIUM=$b timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, d['halos_per_class'])" || exit 1
done
