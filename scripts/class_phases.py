"""Phase clocks (thread 0 of every group, SOAP_ENGINE_PROFILE=1) of ONE class of a workload run alone: WL=c5 python scripts/class_phases.py 2"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
key = os.environ.get("WL", "c5"); k = sys.argv[1]
wl = bench.WORKLOADS[key]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS, float32_masses=os.environ.get("F32_MASSES") == "1" or wl.get("float32_masses", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {kk: v.cpu().numpy() for kk, v in halos.items()}
H = len(hn["index"])
os.environ["SOAP_ENGINE_ONLY_CLASS"] = k
os.environ["SOAP_ENGINE_PROFILE"] = "1"
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({kk: v.copy() for kk, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    try:
        out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    except RuntimeError as e:
        print("run:", e)
ph = eng.phase_cycles()
print("class", k, "halo ms", round(eng.kernel_ms("halo"), 2))
tot = max(ph.get("total", 0), 1)
for n, v in sorted(ph.items(), key=lambda kv: -kv[1]):
    if v > 0: print(f"  {n:24s} {v / tot:6.3f}")
