import csv, glob, sys
rows = []
for f in glob.glob('gpurun_out/tl_s/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:50], r.get('Grid_Size_X', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('k_minmax')]
i0 = starts[-1]
t0 = rows[i0][0]
for s, e, n, g in rows[i0:i0 + 40]:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  grid {g:>9s}  {n}")
    if n.startswith('k_estimate'): break
