#!/bin/bash
for cfg in "256 16384" "512 16384" "1024 16384" "512 8192" "512 32768" "128 16384"; do
  set -- $cfg
  SOAP_ENGINE_NB_SMALL=$1 SOAP_ENGINE_NB_MEDIUM=$2 timeout -k 10 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, d['halos_per_class'])" || exit 1
done
