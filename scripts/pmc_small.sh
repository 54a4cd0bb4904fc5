#!/bin/bash
# PMC passes over the small-halo class alone (C3, SOAP_ENGINE_ONLY_CLASS=3): instruction mix, waits, LDS, memory
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WL=${WL:-c3} CONFIGS=only3
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmcs_$i --output-format csv -- python3 scripts/small_perf.py > $R/gpurun_out/pmcs_$i.log 2>&1) || echo "group $i failed ($grp)"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmcs_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in tot.items():
    if 'k_small' in k:
        print(k)
        for c, v in sorted(d.items()): print(f"   {c:28s} {v:.6g}")
PY
