#!/bin/bash
# Why does a tiny halo take 3.4x longer at 8 wavefront groups per CU than at 2?  PMC counters of the wavefront class alone
# (1e7 particles / 1e4 halos, C3 calculation list, halos below 150 particles) at 512 and at 2048 resident groups.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LAT_MAXNB=150 SOAP_ENGINE_PIPELINE=0
i=0
for nwg in 128 512; do
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  (cd $R && LAT_NWG=$nwg SOAP_ENGINE_PROFILE=0 timeout -k 10 280 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmco_$i --output-format csv -- python3 scripts/latency.py > $R/gpurun_out/pmco_$i.log 2>&1) || echo "group $i failed ($nwg: $grp)"
done
done
