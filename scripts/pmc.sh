#!/bin/bash
# PMC passes over the wavefront class alone (1e7 particles / 1e4 halos, C3 calculation list)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LAT_NWG=512
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  (cd $R && timeout -k 10 280 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_w$i --output-format csv -- python3 scripts/latency.py > $R/gpurun_out/pmc_w$i.log 2>&1) || exit 1
done
