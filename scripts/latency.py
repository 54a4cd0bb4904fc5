"""Per-halo cycle budget of the wavefront class with and without contention (diagnostic)."""
import os, sys
sys.path.insert(0, ".")
os.environ["SOAP_ENGINE_ONLY_CLASS"] = "0"
os.environ.setdefault("SOAP_ENGINE_PROFILE", "1")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = dict(bench.WORKLOADS["c3"], n_part=10_000_000, n_halo=10_000)
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1, device="cuda:0")
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
if os.environ.get("LAT_MAXNB"):
    keep = hn["nr_bound_part"] < int(os.environ["LAT_MAXNB"])
    hn = {k: v[keep] for k, v in hn.items()}
H = len(hn["index"])
for nwg in [int(x) for x in os.environ.get('LAT_NWG', '1,16,512').split(',')]:
    eng = halo_tasks.make_engine(sc, device=0, n_workgroups=nwg)
    for rep in range(2):
        halo_tasks.stage_chunk(eng, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
    n0 = int(eng.kernel_ms("n_wave"))
    ph = eng.phase_cycles()
    print(f"waves={4*nwg} halos={n0} ms={eng.kernel_ms('halo_wave'):.1f}  cycles/halo:",
          {k: int(v / n0) for k, v in ph.items()}, flush=True)
    eng.close()
