"""The longest per-halo chains of a workload (stats column 3: shader clocks spent on the halo by its thread group): who are they?"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS)
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
st = out[4]["stats"]
clk = st[:, 3].astype(np.float64)
order = np.argsort(-clk)[:25]
print("halo kernels %.2f ms; clock unit: 100 MHz counter -> ms = clocks / 1e5" % eng.kernel_ms("halo"))
print("  row   ms(chain)   n_sel    n_cand   nr_bound  central  loops attempts  search_radius_in/out")
for i in order:
    print(f"{i:6d} {clk[i] / 1e5:9.2f} {st[i,1]:9d} {st[i,2]:9d} {hn['nr_bound_part'][i]:9d} {hn['is_central'][i]:6d} {st[i,4]:6d} {st[i,5]:4d}   {hn['search_radius'][i]:.3f} -> {ha['search_radius'][i]:.3f}")
