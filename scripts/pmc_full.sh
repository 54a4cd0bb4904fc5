#!/bin/bash
# PMC passes over the full C3 bench (all three classes concurrently), one counter group per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
i=0
while read -r grp; do
  i=$((i+1))
  (cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_f$i --output-format csv -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_f$i.log 2>&1)
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done <<'EOL'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU
TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA_RD_UNCACHED_32B_sum
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU
EOL
