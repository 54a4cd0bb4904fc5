#!/bin/bash
# C3 halo phase against the number of persistent groups of the 256-thread class (each holds 64 KB of a CU's 160 KB of LDS)
for g1 in ${G1S:-192 256 320 384 448 512}; do
  echo -n "G1=$g1 "; SOAP_ENGINE_G1=$g1 CONFIGS=all python scripts/small_perf.py 2>&1 | grep "^all"
done
