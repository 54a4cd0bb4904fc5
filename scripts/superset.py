"""How much larger than the final sphere is the sorted superset? (diagnostic, SOAP_ENGINE_PROFILE=2)"""
import os, sys
sys.path.insert(0, ".")
os.environ["SOAP_ENGINE_PROFILE"] = os.environ.get("PROF", "2")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
wl = dict(bench.WORKLOADS["c3"], n_part=10_000_000, n_halo=10_000)
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1, device="cuda:0")
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, data)
ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
st = out[4]["stats"]; ok = out[4]["status"] == 0
print("n_loop mean", st[ok, 0].mean(), "n_sel sum", st[ok, 1].sum(), "n_have sum", st[ok, 3].sum(), "n_cand sum", st[ok, 2].sum())
ratio = st[ok, 3] / np.maximum(st[ok, 1], 1)
print("n_have/n_sel percentiles 10/50/90:", np.percentile(ratio, [10, 50, 90]))
final_over_search = out[4]["search_radius"][ok]
for lo, hi in [(0, 64), (64, 256), (256, 1024), (1024, 1 << 30)]:
    m = ok & (st[:, 1] >= lo) & (st[:, 1] < hi)
    print(f"n_sel [{lo},{hi}): halos={m.sum()} mean n_sel={st[m,1].mean():.0f} mean n_have={st[m,3].mean():.0f} loops={st[m,0].mean():.1f}")
