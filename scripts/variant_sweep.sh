#!/bin/bash
# class-alone and whole-phase times of C3 for library variants (scripts/build_variant.sh) and group counts of the 256-thread class
# VARIANTS="default occ4" G1S="512 1024" CONFIGS=only1,all scripts/variant_sweep.sh
for v in ${VARIANTS:-default}; do
  for g1 in ${G1S:-512}; do
    if [ $v = default ]; then unset SOAP_ENGINE_LIB; else export SOAP_ENGINE_LIB=$PWD/soap_amd/lib/variants/libsoap_engine_$v.so; fi
    echo "== variant $v G1=$g1 ${EXTRA_ENV}"
    env $EXTRA_ENV SOAP_ENGINE_G1=$g1 CONFIGS=${CONFIGS:-only1,all} timeout -k 10 300 python scripts/small_perf.py 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
