#!/bin/bash
# C3: upper limit (candidates of the first sphere) of the halos that start in the small-halo class / wavefront class; above it the
# 256-thread class takes them
for nb in ${NBS:-16384 8192 6144 4096 3072 2048}; do
  echo -n "NB_SMALL=$nb "; SOAP_ENGINE_NB_SMALL=$nb CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
done
