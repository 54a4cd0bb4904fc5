"""Per-kernel summary of a rocprofv3 kernel trace (scripts/collect_profile_workload.sh): calls, total and average duration, with
the three thread-group classes of k_halo told apart by their workgroup size and torch's kernels folded into one line."""
import collections, csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
trace = max(glob.glob(f"{src}/stats/*/*kernel_trace.csv"), key=os.path.getmtime)
per = collections.defaultdict(list)
for row in csv.DictReader(open(trace)):
    name = row["Kernel_Name"]
    if "at::native" in name or "rocprim" in name or "rocclr" in name or name.startswith("void "):
        name = "(torch / runtime kernels: data generation, catalogue gather, copies)"
    else:
        name = name.split("(")[0].replace("(anonymous namespace)::", "") or name
        if "k_halo" in row["Kernel_Name"]:
            name = f"k_halo<{row['Workgroup_Size_X']} threads per halo>"
        elif row["Kernel_Name"].startswith("(anonymous namespace)::"):
            name = row["Kernel_Name"][len("(anonymous namespace)::"):].split("(")[0] + f" [{row['Workgroup_Size_X']} threads]"
    per[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
tot = sum(sum(v) for v in per.values())
with open(dst, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / tot, 3), min(v), max(v)])
print(open(dst).read())
