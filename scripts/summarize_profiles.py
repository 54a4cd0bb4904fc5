"""Turn the rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/prof_final) into the summaries
committed under profiles/round1/.  Runs in the build container (no GPU needed)."""
import collections, csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles/round1"
TAG = sys.argv[3] if len(sys.argv) > 3 else "final"
os.makedirs(DST, exist_ok=True)

def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(f"{SRC}/stats/*/*kernel_stats.csv")
shutil.copy(stats, f"{DST}/{TAG}_c3_kernel_stats_all_launches.csv")   # rocprofv3's own summary, every launch


def step_launches(values):
    """(the cleanup launch of the wavefront class -- the halos the small-halo class hands on -- is a launch of its own, shorter
    than the class' main launch: it is counted with the short ones, its bytes are small.)
    bench.py brings the catalogue radii to their steady state before the timed region: one full pass and a
    few passes over the ~35 halos that asked for a larger read radius.  Those short launches of the same
    kernels are not steps: keep the launches within a factor two of the largest one."""
    top = max(values)
    return [v for v in values if v >= 0.5 * top]


# per-kernel durations of the step launches, from the kernel trace of the same run
trace = newest(f"{SRC}/stats/*/*kernel_trace.csv")
def is_halo(name):
    """the kernels of the halo phase: the thread-group classes and the small-halo class"""
    return "k_halo" in name or "k_small" in name


def kname(row, wg_key):
    """the three thread-group classes are one kernel name compiled three times: tell them apart by the workgroup size"""
    name = row["Kernel_Name"]
    if "k_halo" in name:
        return f"k_halo<{row[wg_key]} threads per halo>"
    if "k_small" in name:
        return name[name.index("k_small"):].split("(")[0]
    return name


per = collections.defaultdict(list)
for row in csv.DictReader(open(trace)):
    per[kname(row, "Workgroup_Size_X")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
rows = []
for name, d in per.items():
    keep = step_launches(d)
    rows.append((name, len(keep), sum(keep), sum(keep) / len(keep), min(keep), max(keep), len(d) - len(keep)))
tot = sum(r[2] for r in rows)
with open(f"{DST}/{TAG}_c3_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "ShortLaunchesExcluded"])
    for r in sorted(rows, key=lambda r: -r[2]):
        w.writerow([r[0], r[1], r[2], r[3], round(100.0 * r[2] / tot, 4), r[4], r[5], r[6]])
dur = {r[0]: r[3] * 1e-6 for r in rows if is_halo(r[0])}

def pmc(name, counter):
    f = newest(f"{SRC}/{name}/*/*counter_collection.csv")
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and is_halo(row["Kernel_Name"]):
            vals[kname(row, "Workgroup_Size")].append(float(row["Counter_Value"]))
    return {k: sum(step_launches(v)) / len(step_launches(v)) for k, v in vals.items()}

fetch_kb, write_kb = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
line = json.loads(open(f"{SRC}/bench_c3.json").read().strip().splitlines()[-1])
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --workload c3 --steps 1 --warmup 0 --passes-per-step 4 --no-cpu-baseline --no-full-set (scripts/collect_profiles.sh; one counter per run)",
    "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB per launch (rocprofv3 derived counters).  MI355X_MICROARCH.md: on gfx950 "
                 "FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled ('corrected'); the factor is calibrated "
                 "for 16 B/lane streaming reads, our loads are 8 B/lane and partly gathers, so raw and corrected bracket the truth.",
    "per_kernel": {k: {"fetch_KiB": fetch_kb.get(k), "write_KiB": write_kb.get(k), "avg_ms": dur.get(k)} for k in sorted(set(fetch_kb) | set(write_kb))},
    "workload": line["config"]["workload"], "workload_key": line["config"].get("workload_key", "c3"), "seed": line["config"]["seed"],
}
raw = (sum(fetch_kb.values()) + sum(write_kb.values())) * 1024.0
cor = (2.0 * sum(fetch_kb.values()) + sum(write_kb.values())) * 1024.0
out["traffic_bytes_per_step_raw"] = raw
out["traffic_bytes_per_step_corrected"] = cor
out["algorithmic_bytes_per_step"] = line["roofline"]["algorithmic_bytes"]
json.dump(out, open(f"{DST}/{TAG}_c3_hbm_traffic.json", "w"), indent=1)
line["roofline"]["traffic"] = cor   # the value measured in THIS profile set (bench.py quotes the committed one)
json.dump(line, open(f"{DST}/{TAG}_c3_bench.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])

# ---- SQ / L2 counters of the halo kernels (scripts/collect_profiles.sh passes sq1..), step launches only ----------------
sq = {}
for d in sorted(glob.glob(f"{SRC}/sq*/")):
    files = glob.glob(f"{d}/*/*counter_collection.csv")
    if not files:
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if is_halo(row["Kernel_Name"]):
            vals[kname(row, "Workgroup_Size")][row["Counter_Name"]].append(
                (float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), int(row["VGPR_Count"]),
                 int(row["Accum_VGPR_Count"]), int(row["SGPR_Count"]), int(row["Scratch_Size"]), int(row["LDS_Block_Size"]),
                 int(row["Workgroup_Size"]), int(row["Grid_Size"])))
    for k, cs in vals.items():
        for c, lst in cs.items():
            top = max(x[1] for x in lst)
            keep = [x for x in lst if x[1] >= 0.5 * top]   # the step launches (see step_launches)
            e = sq.setdefault(k, {"counters": {}})
            e["counters"][c] = sum(x[0] for x in keep) / len(keep)
            e.update(vgpr=keep[0][2], agpr=keep[0][3], sgpr=keep[0][4], scratch_bytes_per_lane=keep[0][5], lds_bytes_per_group=keep[0][6],
                     workgroup=keep[0][7], grid=keep[0][8], launches_averaged=len(keep))
if sq:
    for k, e in sq.items():
        c = e["counters"]
        d = {}
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            d["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_ANY" in c:
            d["active_inst_fraction_of_wave_cycles"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_INST_ANY" in c:
            d["wait_inst_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        insts = {n: c[n] for n in c if n.startswith("SQ_INSTS_")}
        if insts:
            d["instruction_mix"] = {n: v / sum(insts.values()) for n, v in insts.items()}
        e["derived"] = d
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --workload c3 --steps 1 --warmup 0 --passes-per-step 4 "
                          "--no-cpu-baseline --no-full-set (scripts/collect_profiles.sh; one counter group per run; the three classes of k_halo run "
                          "concurrently, counters are per dispatch, averaged over the full-size launches)",
               "per_kernel": sq}, open(f"{DST}/{TAG}_c3_sq_counters.json", "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in sq.items()}, indent=1))
