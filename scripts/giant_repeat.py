"""Two identical runs of a workload with the engine's trace on: do the same halos take the same path?"""
import os, sys
os.environ["SOAP_ENGINE_TRACE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "giant"]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
e = halo_tasks.make_engine(sc, device=0)
tabs = []
for rep in range(3):
    print("=== run", rep, file=sys.stderr, flush=True)
    halo_tasks.stage_chunk(e, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.process_halos(e, props, sc, ha, None, unpack=False)
    tabs.append(e.fetch(table=True)["table"].copy())
for rep in (1, 2):
    d = np.flatnonzero((tabs[rep] != tabs[0]).any(axis=1))
    print("rows that differ between run 0 and run", rep, ":", d[:20], len(d))
