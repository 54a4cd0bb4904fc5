#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the LAST pass of scripts/ab.py for one configuration: who runs when.
#   scripts/timeline_ab.sh <tag> [ENV=value,ENV=value | -]      (WL / SEED / REPS as for ab.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-tl}; CFG=${2:--}
rm -rf $R/gpurun_out/tl_$TAG
(cd $R && REPS=${REPS:-2} timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$TAG --output-format csv -- python3 scripts/ab.py "$CFG" > $R/gpurun_out/tl_$TAG.log 2>&1) || echo failed
cd $R && python3 - $TAG <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f'gpurun_out/tl_{tag}/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''), r.get('LDS_Block_Size', '')))
rows.sort()
# a run starts with k_estimate (+ k_predict); the giants' pipeline calls k_estimate again inside a run (with a rebuild radius): the
# last k_predict marks the last run where there is one
pred = [i for i, r in enumerate(rows) if r[2].startswith('k_predict')]
starts = pred if pred else [i for i, r in enumerate(rows) if r[2].startswith('k_estimate')]
i0 = starts[-1]
while i0 > 0 and (rows[i0 - 1][2].startswith('k_predict') or rows[i0 - 1][2].startswith('k_estimate')):
    i0 -= 1
t0 = rows[i0][0]
with open(f'gpurun_out/tl_{tag}.txt', 'w') as out:
    for s, e, n, g, w, l in rows[i0:]:
        line = f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  grid {g:>8s} wg {w:>5s} lds {l:>6s}  {n}"
        print(line); out.write(line + "\n")
PY
rm -rf $R/gpurun_out/tl_$TAG
