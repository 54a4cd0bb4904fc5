#!/bin/bash
# kernel timeline of the last pass of `bench.py $ARGS` (rocprofv3 --kernel-trace): who runs when
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${ARGS:---workload c3u}
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/tlb --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --no-full-set --steps 1 --warmup 0 --passes-per-step 2 > $R/gpurun_out/tlb.log 2>&1) || echo failed
cd $R && python3 - <<'PY'
import csv, glob, os
rows = []
f = max(glob.glob('gpurun_out/tlb/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:48], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Workgroup_Size_X', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_minmax' in r[2] or 'k_cellid' in r[2]]
# the last staging marks the start of the last pass
i0 = starts[-1]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][0] < 2e6 and ('k_minmax' in rows[i0-1][2] or 'k_cellid' in rows[i0-1][2]): i0 -= 1
t0 = rows[i0][0]
for s, e, n, g, w in rows[i0:]:
    if (e - s) < 100e3: continue
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  grid {g:>8s} wg {w:>5s}  {n}")
PY
