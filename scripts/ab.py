"""A/B of engine switches on one workload (diagnostic): the box is generated once, every configuration -- a comma-separated list
of ENV=value given on the command line, "-" for the defaults -- gets a fresh engine, one untimed and REPS timed passes
(stage + process_halos, every halo finished inside the resident box).  Prints the halo phase and the class clocks per
configuration; CHECK=1 also compares every configuration's table with the first one's."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

key = os.environ.get("WL", "c3")
seed = int(os.environ.get("SEED", "1234"))
reps = int(os.environ.get("REPS", "3"))
check = os.environ.get("CHECK") == "1"
wl = bench.WORKLOADS[key]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=seed, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS, float32_masses=os.environ.get("F32_MASSES") == "1" or wl.get("float32_masses", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl, inertia=os.environ.get("BENCH_INERTIA") == "1")
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
ref = None
names = ("reorder", "halo", "halo_small", "halo_small512", "halo_small1024", "halo_wave", "halo_wg256", "halo_wg1024", "halo_team", "pipe_build", "pipe_stage_a", "pipe_stage_b", "run_classify", "run_total")
counts = ("n_small128", "n_small512", "n_small1024", "n_small_escalated", "n_wave", "n_wg256", "n_wg1024", "n_pipeline", "n_passes", "n_predicted", "n_team", "n_team_failed")
for cfg in sys.argv[1:] or ["-"]:
    sets = [kv.split("=", 1) for kv in cfg.split(",") if "=" in kv]
    for k, v in sets:
        os.environ[k] = v
    try:
        eng = halo_tasks.make_engine(sc, device=0)
        ms = {n: [] for n in names}
        for rep in range(reps + 1):
            halo_tasks.stage_chunk(eng, data)
            ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
            out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
            if rep:
                for n in names:
                    ms[n].append(eng.kernel_ms(n))
        line = " ".join(f"{n.replace('halo_', '')} {np.median(v):6.2f}" for n, v in ms.items())
        cnt = " ".join(f"{n[2:]} {int(eng.kernel_ms(n))}" for n in counts)
        left = int(np.sum(ha["done"] == 0))
        msg = ""
        if check:
            t = eng.fetch(table=True)["table"]
            if ref is None:
                ref = (t, out[4]["status"].copy(), out[4]["stats"][:, [0, 1, 2, 4, 5]].copy(), ha["search_radius"].copy())
            else:
                scale = np.maximum(np.abs(ref[0]).max(axis=0, keepdims=True), 1e-300)
                bad = np.abs(t - ref[0]) > 1e-11 * np.abs(ref[0]) + 1e-13 * scale
                if bad.any() and os.environ.get("CHECK_COLUMNS") == "1":
                    # which columns: (calculation, column within it, count, worst relative difference)
                    ncols = [eng.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)]
                    offs = np.concatenate([[0], np.cumsum(ncols)])
                    for j in np.flatnonzero(bad.any(axis=0))[:40]:
                        c = int(np.searchsorted(offs, j, side="right") - 1)
                        rel = np.abs(t[:, j] - ref[0][:, j]) / np.maximum(np.abs(ref[0][:, j]), 1e-300)
                        print(f"   column {j}: calculation {c} ({type(props[c]).__name__}) + {j - offs[c]}: {int(bad[:, j].sum())} rows, worst rel {rel[bad[:, j]].max():.3g}", flush=True)
                msg = f" | vs first: values differing {int(bad.sum())} status {int(np.sum(out[4]['status'] != ref[1]))} stats {int(np.sum(out[4]['stats'][:, [0, 1, 2, 4, 5]] != ref[2]))} radii {int(np.sum(ha['search_radius'] != ref[3]))}"
        print(f"{cfg:40s} {line} | {cnt} left {left}{msg}", flush=True)
        eng.close()
    finally:
        for k, v in sets:
            os.environ.pop(k, None)
