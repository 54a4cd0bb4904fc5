import sys; sys.path.insert(0,'.')
import numpy as np, bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
wl = bench.WORKLOADS["c3"]
for f in (8.0, 16.0, 32.0):
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", read_radius_factor=f)
    cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
    hn = {k: v.cpu().numpy() for k, v in halos.items()}
    eng = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(eng, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(len(hn["index"]), dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
    print(f, "unfinished", out[2], "max read", hn["read_radius"].max(), flush=True)
    eng.close()
