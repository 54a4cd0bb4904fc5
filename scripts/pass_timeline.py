"""Where one pass spends its wall-clock time: host-side breakdown with a device synchronisation after every step, and the
engine calls inside process_halos timed one by one (diagnostic; WL as for bench.py)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.catalogue import spatial_sort_order_device
from soap_amd.core.result_set import PackedLayout
from soap_amd.core import halo_tasks
from soap_amd import engine as engine_mod
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda:0")
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device=dev, max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS)
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
engine = halo_tasks.make_engine(sc, device=0)
d_cofp = torch.as_tensor(hn["cofp"], device=dev); d_index = torch.as_tensor(hn["index"], device=dev)
cell = sc.boxsize / bench.TOPCELLS
packed = PackedLayout(props, engine.desc_columns)
inner = {}
def timed(name):
    real = getattr(engine_mod.Engine, name)
    def wrap(self, *a, **k):
        t = time.perf_counter(); r = real(self, *a, **k); inner[name] = inner.get(name, 0.0) + (time.perf_counter() - t) * 1e3; return r
    setattr(engine_mod.Engine, name, wrap)
for nm in ("stage_halos", "run", "fetch", "_fetch_small", "stage_particles"):
    timed(nm)
def lap(t, name, acc):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t) * 1e3; return t1
for rep in range(6):
    acc = {}; inner.clear()
    torch.cuda.synchronize(); t = t00 = time.perf_counter()
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    t = lap(t, "copy halo arrays", acc)
    halo_tasks.stage_chunk(engine, data)
    t = lap(t, "stage_chunk", acc)
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    t = lap(t, "process_halos", acc)
    o = spatial_sort_order_device(d_cofp, d_index, np.array([cell] * 3), (bench.TOPCELLS,) * 3)
    t = lap(t, "catalogue order", acc)
    narrow, wide = packed.pack_engine(engine, dev, order=o)
    cat = (narrow, wide, d_cofp[o], d_index[o])
    t = lap(t, "pack in order", acc)
    acc["total"] = (t - t00) * 1e3
    km = {k: engine.kernel_ms(k) for k in ("mesh", "reorder", "halo", "run_setup", "run_pre", "run_classify", "run_total", "stage_host0", "stage_host1", "stage_host2", "stage_host3")}
    print(rep, {k: round(v, 2) for k, v in acc.items()}, "| inside:", {k: round(v, 2) for k, v in inner.items()}, "|", {k: round(v, 2) for k, v in km.items()}, flush=True)
