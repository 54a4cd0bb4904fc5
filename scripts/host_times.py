"""Wall-clock split of one C3 step on the host side (diagnostic)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
eng = halo_tasks.make_engine(sc, device=0)
calcs = [hp.calc_desc(props, i) for i, hp in enumerate(props)]
td = halo_tasks.target_density_of(props, sc)
for rep in range(4):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    halo_tasks.stage_chunk(eng, data); t.append(time.perf_counter())
    eng.stage_halos(hn["cofp"], hn["search_radius"], hn["read_radius"], hn["index"], hn["is_central"], hn["nr_bound_part"]); t.append(time.perf_counter())
    eng.run(calcs, td); t.append(time.perf_counter())
    out = eng.fetch(table=False); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print(f"stage_chunk {d[0]:.1f}  stage_halos {d[1]:.1f}  run {d[2]:.1f}  fetch {d[3]:.1f}  | kernels: mesh {eng.kernel_ms('mesh'):.1f} reorder {eng.kernel_ms('reorder'):.1f} halo {eng.kernel_ms('halo'):.1f} | run: pre {eng.kernel_ms('run_pre'):.1f} setup {eng.kernel_ms('run_setup'):.1f} classify {eng.kernel_ms('run_classify'):.1f} total {eng.kernel_ms('run_total'):.1f}", flush=True)
