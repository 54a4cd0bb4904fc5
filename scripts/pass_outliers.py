"""Which step of a C3 pass produces the occasional long pass?"""
import os, sys, time, gc
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import halo_tasks
from soap_amd.core.halo_tasks import target_density_of
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda:0")
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device=dev)
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
engine = halo_tasks.make_engine(sc, device=0)
calcs = [hp.calc_desc(props, i) for i, hp in enumerate(props)]
td = target_density_of(props, sc)
rows = []
for rep in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    halo_tasks.stage_chunk(engine, data)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    engine.stage_halos(hn["cofp"], hn["search_radius"], hn["read_radius"], hn["index"], hn["is_central"], hn["nr_bound_part"],
                       resident_radius=np.full(H, np.inf))
    t2 = time.perf_counter()
    engine.run(calcs, td)
    t3 = time.perf_counter()
    out = engine.fetch(table=False)
    t4 = time.perf_counter()
    rows.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), engine.kernel_ms("halo"), engine.kernel_ms("run_total"),
                 engine.kernel_ms("run_setup"), engine.kernel_ms("run_classify"), engine.kernel_ms("halo_wave"), engine.kernel_ms("halo_wg256"), engine.kernel_ms("halo_wg1024")])
r = np.array(rows)
print("columns: stage_chunk stage_halos run fetch | halo run_total run_setup run_classify wave wg256 wg1024")
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print(r[2:])
print("median", np.median(r[2:], axis=0)); print("mean", r[2:].mean(axis=0)); print("max", r[2:].max(axis=0))
