#!/bin/bash
# C3 halo phase: small-halo kernels launched before the thread-group classes, with fewer persistent waves
echo -n "default            "; CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
for gs in ${GSS:-512 768 1024 1536 2816}; do
  echo -n "SMALL_FIRST GS=$gs "; SOAP_ENGINE_SMALL_FIRST=1 SOAP_ENGINE_GS=$gs CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
done
