#!/bin/bash
# C3 with the inertia tensors on: which halos of the wavefront class start in the small-halo kernels (SOAP_ENGINE_SMALL)
for m in 4 1 0; do
  echo "SOAP_ENGINE_SMALL=$m"
  BENCH_INERTIA=1 SOAP_ENGINE_SMALL=$m CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all\|^nosmall"
done
