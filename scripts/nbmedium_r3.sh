#!/bin/bash
# C3: upper limit (candidates) of the 256-thread class; above it 1024 threads per halo
for nm in ${NMS:-262144 230000 200000 180000 160000 131072}; do
  echo -n "NB_MEDIUM=$nm "; SOAP_ENGINE_NB_MEDIUM=$nm CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
done
