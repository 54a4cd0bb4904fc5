"""Which halos cost what: per-halo shader clocks (stats[:, 3]) of a C3 step, by class and size (diagnostic)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
st = out[4]["stats"]
cyc = st[:, 3].astype(np.float64)
print("kernel ms", {k: round(eng.kernel_ms(k), 1) for k in ("halo", "halo_wave", "halo_wg256", "halo_wg1024")})
order = np.argsort(-cyc)
print("top halos by clocks: (ms @2.4GHz, n_loop, n_sel, n_cand, nr_bound)")
for i in order[:25]:
    print(f"  {cyc[i]/2.4e6:8.2f} ms  loops={st[i,0]} n_sel={st[i,1]} n_cand={st[i,2]} nb={hn['nr_bound_part'][i]} status={out[4]['status'][i]}")
# by n_sel bins
bins = [0, 64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 22]
for lo, hi in zip(bins[:-1], bins[1:]):
    m = (st[:, 1] >= lo) & (st[:, 1] < hi)
    if m.any():
        print(f"n_sel [{lo},{hi}): halos={m.sum()} sum={cyc[m].sum()/2.4e6:.1f} WG-ms mean={cyc[m].mean()/2.4e3:.1f} us max={cyc[m].max()/2.4e3:.1f} us")
