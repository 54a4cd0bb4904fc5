"""Per-halo time of the property stage (stats column 3: shader clocks of the group that ran the halo's calculations over
all particles) for the largest halos of a workload."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "giant"]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", max_halo=wl.get("max_halo"), hydro=wl.get("hydro", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
e = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(e, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(e, props, sc, ha, None, unpack=False, resident_radius=np.inf)
st = out[4]["stats"]
order = np.argsort(-st[:, 3])[:12]
for i in order:
    print("row %5d n_sel %8d n_cand %9d loops %2d cycles %.3g" % (i, st[i, 1], st[i, 2], st[i, 0], st[i, 3]))
print("halo ms", e.kernel_ms("halo"), "B", e.kernel_ms("pipe_stage_b"), "build", e.kernel_ms("pipe_build"))
