#!/bin/bash
# kernel timeline of the last pass of ONE rank's share of the C3 box (scripts/rank_share.py under rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RANKS=${RANKS:-1}
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tlr --output-format csv -- python3 scripts/rank_share.py ${N:-8} > $R/gpurun_out/tlr.log 2>&1) || echo failed
cd $R && grep "^rank\|^N=" gpurun_out/tlr.log | cut -c1-400; python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/tlr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Workgroup_Size_X', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_estimate' in r[2]]
i0 = starts[-1]
# back up to the mesh build of the same pass
j0 = i0
while j0 > 0 and rows[i0][0] - rows[j0 - 1][0] < 4e6: j0 -= 1
t0 = rows[j0][0]
for s, e, n, g, w in rows[j0:]:
    if (e - s) < 20e3 and 'k_' not in n: continue
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  grid {g:>8s} wg {w:>5s}  {n}")
PY
