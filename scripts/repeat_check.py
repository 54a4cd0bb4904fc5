"""Does a second run over the same staged chunk reproduce the table bit for bit?  (diagnostic; WL / SEED as for ab.py; every
command-line argument is a configuration "ENV=value,ENV=value" or "-")"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=int(os.environ.get("SEED", "1234")), device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
for cfg in sys.argv[1:] or ["-"]:
    sets = [kv.split("=", 1) for kv in cfg.split(",") if "=" in kv]
    for k, v in sets:
        os.environ[k] = v
    eng = halo_tasks.make_engine(sc, device=0)
    tabs = []
    for rep in range(3):
        halo_tasks.stage_chunk(eng, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
        tabs.append((eng.fetch(table=True)["table"], out[4]["stats"].copy()))
    for rep in (1, 2):
        bad = np.argwhere(tabs[rep][0] != tabs[0][0])
        rows = np.unique(bad[:, 0])
        print(f"{cfg}: run {rep} vs run 0: {len(bad)} values in {len(rows)} rows differ; rows {rows[:8]} n_sel {tabs[0][1][rows[:8], 1]} nr_bound {hn['nr_bound_part'][rows[:8]]} cols {np.unique(bad[:, 1])[:12]}", flush=True)
    eng.close()
    for k, v in sets:
        os.environ.pop(k, None)
