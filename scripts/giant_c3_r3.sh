#!/bin/bash
# C3 with its largest halos through the multi-workgroup pipeline (SOAP_ENGINE_NB_GIANT: halos above this many candidates)
for ng in 0 400000 262144 131072 65536 32768; do
  echo "NB_GIANT=$ng"
  if [ $ng = 0 ]; then CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"; else SOAP_ENGINE_NB_GIANT=$ng CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"; fi
done
