"""Wall time of every pass of a bench-like loop and of the engine's read-back inside it (diagnostic): how often is the copy of
statuses / radii / stats to the host slow?  python scripts/fetch_times.py [passes]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import halo_tasks
from soap_amd import engine as engine_mod
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS, float32_masses=wl.get("float32_masses", False))
props = bench.build_props(sc, wl)[4]
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
engine = halo_tasks.make_engine(sc, device=0)
inner = {}
for nm in ("stage_halos", "run", "_fetch_small", "stage_particles"):
    real = getattr(engine_mod.Engine, nm)
    def wrap(self, *a, _real=real, _nm=nm, **k):
        t = time.perf_counter(); r = _real(self, *a, **k); inner[_nm] = inner.get(_nm, 0.0) + (time.perf_counter() - t) * 1e3; return r
    setattr(engine_mod.Engine, nm, wrap)
rows = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    inner.clear()
    t0 = time.perf_counter()
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.stage_chunk(engine, data)
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    torch.cuda.synchronize()
    rows.append([(time.perf_counter() - t0) * 1e3] + [inner.get(k, 0.0) for k in ("stage_particles", "stage_halos", "run", "_fetch_small")])
a = np.array(rows[2:])
print("columns: pass, stage_particles, stage_halos, run, fetch  [ms]")
print("median", np.round(np.median(a, axis=0), 2), "mean", np.round(a.mean(axis=0), 2), "max", np.round(a.max(axis=0), 2))
print("fetch per pass:", np.round(a[:, 4], 1))
