#!/bin/bash
# dynamic instruction counts of the small-halo class by phase: the kernel is cut short after phase k (SOAP_ENGINE_PROFILE=100+k)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WL=${WL:-c3} CONFIGS=only3
for k in 8 1 2 3 4 5 6 7; do
  (cd $R && SOAP_ENGINE_PROFILE=$((100+k)) timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmcp_$k --output-format csv -- python3 scripts/small_perf.py > $R/gpurun_out/pmcp_$k.log 2>&1) || echo "stop $k failed"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
names = {8: "header + row zero", 1: "+ first build (select, gather, sort, cumsum)", 2: "+ first sphere count", 3: "+ radius loop (stage A)", 4: "+ moments (all)", 5: "+ scan (all)", 6: "+ flow rates", 7: "everything"}
prev = None
for k in (8, 1, 2, 3, 4, 5, 6, 7):
    d = collections.defaultdict(float)
    for f in glob.glob(f'gpurun_out/pmcp_{k}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_small' in r['Kernel_Name']: d[r['Counter_Name']] += float(r['Counter_Value'])
    n = 4 * 53474.0   # four runs of 53474 halos
    row = {c: v / n for c, v in d.items()}
    print(f"stop {k} {names[k]:46s}", "  ".join(f"{c.replace('SQ_','').replace('INSTS_','')}: {row.get(c,0):8.0f}" for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_BRANCH","SQ_INSTS_VMEM_RD","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY")))
PY
