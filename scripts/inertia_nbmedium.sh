#!/bin/bash
# C3 with the inertia tensors on: upper limit of the 256-thread class (candidates); above it 1024 threads per halo
for nm in ${NMS:-262144 131072 65536 32768}; do
  echo -n "NB_MEDIUM=$nm "; BENCH_INERTIA=1 SOAP_ENGINE_NB_MEDIUM=$nm CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
done
