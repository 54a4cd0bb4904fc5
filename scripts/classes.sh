#!/bin/bash
# per-class phase profile (each class alone on the GPU); WL=c3|c5, CLASSES="0 1 2"
for k in ${CLASSES:-0 1 2}; do
  SOAP_ENGINE_PROFILE=1 SOAP_ENGINE_ONLY_CLASS=$k timeout -k 10 300 python bench.py --workload ${WL:-c3} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['phase_cycles']['total']; print('class $k', {k:round(v,1) for k,v in d['kernel_ms'].items() if k.startswith('halo')}, d['halos_per_class'], {k:round(v/t,3) for k,v in d['phase_cycles'].items() if v}, 'total_cycles %.3g'%t)" || exit 1
done
