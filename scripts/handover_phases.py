"""Phase clocks of the wavefront class on the population the small-halo class hands on (diagnostic): the halos of the box with more bound
particles than the 512-entry slab takes (NB_LO, default 340 = 512 / 1.5) and at most NB_HI, run alone with SOAP_ENGINE_PROFILE=1 (which
keeps them on the general path).  WL as for bench.py."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
key = os.environ.get("WL", "c3")
wl = bench.WORKLOADS[key]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS, float32_masses=wl.get("float32_masses", False))
props = bench.build_props(sc, wl)[4]
hn = {kk: v.cpu().numpy() for kk, v in halos.items()}
lo, hi = int(os.environ.get("NB_LO", "340")), int(os.environ.get("NB_HI", "4000"))
keep = np.flatnonzero((hn["nr_bound_part"] > lo) & (hn["nr_bound_part"] <= hi))
hn = {kk: v[keep] for kk, v in hn.items()}
H = len(keep)
os.environ["SOAP_ENGINE_ONLY_CLASS"] = "0"
os.environ["SOAP_ENGINE_PROFILE"] = "1"
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({kk: v.copy() for kk, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
ph = eng.phase_cycles()
st = out[4]["stats"]
print(f"{H} halos, {int(np.sum(ha['done'] == 1))} done in class 0, halo ms {eng.kernel_ms('halo'):.2f}, mean selected {st[:, 1].mean():.0f} candidates {st[:, 2].mean():.0f} rounds {st[:, 0].mean():.1f}")
tot = max(ph.get("total", 0), 1)
for n, v in sorted(ph.items(), key=lambda kv: -kv[1]):
    if v > 0: print(f"  {n:24s} {v / tot:6.3f}")
