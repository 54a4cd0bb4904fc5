#!/bin/bash
# kernel timeline of the last process_halos call of a workload (rocprofv3 --kernel-trace): who runs when
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WL=${WL:-c3} CONFIGS=${CONFIGS:-all}
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tl --output-format csv -- python3 scripts/small_perf.py > $R/gpurun_out/tl.log 2>&1) || echo failed
cd $R && python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('LDS_Block_Size','')))
rows.sort()
# last k_estimate marks the start of the last run
starts = [i for i, r in enumerate(rows) if 'k_estimate' in r[2]]
i0 = starts[-1]
t0 = rows[i0][0]
for s, e, n, g, l in rows[i0:]:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  grid {g:>8s} lds {l:>6s}  {n}")
PY
