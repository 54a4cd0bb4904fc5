import csv, glob, collections, os
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmco_*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_halo" in r["Kernel_Name"] and r["Workgroup_Size"] == "64"]
    if not rows: continue
    # the last launch of the class kernel (second repetition)
    last = max(int(r["Dispatch_Id"]) for r in rows)
    grid = None
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            grid = int(r["Grid_Size"]); res[grid][r["Counter_Name"]] = float(r["Counter_Value"]); res[grid]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
gs = sorted(res)
print("resident wavefront groups:", [g // 64 for g in gs])
for c in sorted(set().union(*[set(res[g]) for g in gs])):
    vals = [res[g].get(c) for g in gs]
    ratio = (vals[-1] / vals[0]) if all(v for v in vals) else None
    print(f"{c:32s}", " ".join(f"{v:14.4g}" if v is not None else "             -" for v in vals), f"  x{ratio:.2f}" if ratio else "")
