"""Per-halo facts of a workload for offline work on schedules and size predictors (diagnostic): catalogue columns going in,
radii coming out, stats columns of a run with the per-halo shader clocks (column 3) and of a run with the size of the
sorted superset there (SOAP_ENGINE_PROFILE=2).  Writes gpurun_out/halo_dump_<workload>_<seed>.npz."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

key = os.environ.get("WL", "c3")
seed = int(os.environ.get("SEED", "1234"))
wl = bench.WORKLOADS[key]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=seed, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS)
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
out = {}
for tag, prof in (("clk", None), ("have", "2")):
    if prof is None:
        os.environ.pop("SOAP_ENGINE_PROFILE", None)
    else:
        os.environ["SOAP_ENGINE_PROFILE"] = prof
    eng = halo_tasks.make_engine(sc, device=0)
    for rep in range(2):
        halo_tasks.stage_chunk(eng, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        res = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    out["stats_" + tag] = res[4]["stats"].copy()
    out["status_" + tag] = res[4]["status"].copy()
    out["search_out"] = ha["search_radius"].copy()
    out["read_out"] = ha["read_radius"].copy()
    print(tag, "halo kernels %.2f ms" % eng.kernel_ms("halo"), {k: round(eng.kernel_ms(k), 2) for k in ("halo_wave", "halo_wg256", "halo_wg1024", "halo_small", "halo_small512")},
          {k: int(eng.kernel_ms(k)) for k in ("n_wave", "n_wg256", "n_wg1024", "n_small128", "n_small512", "n_small_escalated")})
    del eng
np.savez_compressed("gpurun_out/halo_dump_%s_%d.npz" % (key, seed), boxsize=sc.boxsize, m_p=float(data[1]["Masses"][0].item()),
                    target_density=float(halo_tasks.target_density_of(props, sc)), critical_density=float(sc.critical_density), a=float(sc.a),
                    **{k: v for k, v in hn.items()}, **out)
print("saved", H)
