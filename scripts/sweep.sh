#!/bin/bash
# threshold sweep of the class boundaries (estimated sphere content) on the C3 workload
for cfg in ${CFGS:-"8192 131072" "6144 131072" "4096 131072" "3072 131072" "2048 131072" "4096 65536"}; do
  set -- $cfg
  SOAP_ENGINE_NB_SMALL=$1 SOAP_ENGINE_NB_MEDIUM=$2 timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, d['halos_per_class'])" || exit 1
done
