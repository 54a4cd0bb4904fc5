"""Where one C3 pass spends its wall-clock time: host-side breakdown with a device synchronisation after every step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.catalogue import gather_catalogue_packed
from soap_amd.core.result_set import PackedLayout
from soap_amd.core import chunk_tasks, halo_tasks
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
table = None
def lap(t, name, acc):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t) * 1e3; return t1
for rep in range(6):
    acc = {}
    torch.cuda.synchronize(); t = t00 = time.perf_counter()
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    t = lap(t, "copy halo arrays", acc)
    halo_tasks.stage_chunk(engine, data)
    t = lap(t, "stage_chunk", acc)
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=np.inf)
    t = lap(t, "process_halos", acc)
    raw = out[4]; okm = raw["status"] == 0
    stats, status = raw["stats"], raw["status"]
    t = lap(t, "host bookkeeping", acc)
    if table is None:
        table = torch.zeros((H, engine.row_columns), dtype=torch.float64, device=dev)
    engine.copy_table_to(table.view(-1))
    t = lap(t, "copy_table", acc)
    narrow, wide = packed.pack(table)
    t = lap(t, "pack", acc)
    cat = gather_catalogue_packed(narrow, wide, d_cofp, d_index, np.array([cell] * 3), (bench.TOPCELLS,) * 3)
    t = lap(t, "gather_catalogue", acc)
    acc["total"] = (t - t00) * 1e3
    km = {k: engine.kernel_ms(k) for k in ("mesh", "reorder", "halo", "run_setup", "run_pre", "run_classify", "run_total")}
    print(rep, {k: round(v, 2) for k, v in acc.items()}, {k: round(v, 2) for k, v in km.items()}, flush=True)
