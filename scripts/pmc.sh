#!/bin/bash
# PMC passes over the wavefront class alone (1e7 particles / 1e4 halos, C3 calculation list, tiny halos only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LAT_MAXNB=150
i=0
for nwg in 64 512; do
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  (cd $R && LAT_NWG=$nwg SOAP_ENGINE_PROFILE=0 timeout -k 10 280 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_i$i --output-format csv -- python3 scripts/latency.py > $R/gpurun_out/pmc_i$i.log 2>&1) || exit 1
done
done
