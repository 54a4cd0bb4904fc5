"""Sizes of the halos of a workload as the engine sees them (diagnostic): sorted superset (n_have, SOAP_ENGINE_PROFILE=2),
final sphere (n_sel), candidates (n_cand), radius rounds -- what a small-halo kernel with the slab in LDS has to hold."""
import os, sys
sys.path.insert(0, ".")
os.environ["SOAP_ENGINE_PROFILE"] = "2"
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, data)
ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
st = out[4]["stats"]; ok = out[4]["status"] == 0
print("ok", ok.sum(), "of", H, "n_loop mean", st[ok, 0].mean(), "attempts max", st[ok, 5].max())
n_sel, n_cand, n_have, loops = st[:, 1], st[:, 2], st[:, 3], st[:, 4]
print("totals: n_sel %.4g n_have %.4g n_cand %.4g" % (n_sel[ok].sum(), n_have[ok].sum(), n_cand[ok].sum()))
bins = [0, 32, 64, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 65536, 1 << 30]
print("by n_have (sorted superset):  halos  cum%%   sum n_have  sum n_sel  sum n_cand (share)  mean loops")
cum = 0
for lo, hi in zip(bins[:-1], bins[1:]):
    m = ok & (n_have >= lo) & (n_have < hi)
    cum += m.sum()
    if m.any():
        print(f"  [{lo:6d},{hi:10d}) {m.sum():7d} {100.0*cum/ok.sum():6.2f}  {n_have[m].sum():11d} {n_sel[m].sum():11d} {n_cand[m].sum():12d} ({100.0*n_cand[m].sum()/n_cand[ok].sum():5.1f}%)  {loops[m].mean():.2f}")
print("n_cand percentiles 10/50/90/99:", np.percentile(n_cand[ok], [10, 50, 90, 99]))
print("n_have/n_sel percentiles 10/50/90:", np.percentile(n_have[ok] / np.maximum(n_sel[ok], 1), [10, 50, 90]))
print("nr_bound percentiles 10/50/90/99:", np.percentile(hn["nr_bound_part"], [10, 50, 90, 99]))
np.savez_compressed("gpurun_out/size_hist_%s.npz" % os.environ.get("WL", "c3"), n_sel=n_sel, n_cand=n_cand, n_have=n_have, loops=loops, ok=ok,
                    nr_bound=hn["nr_bound_part"], search_radius=hn["search_radius"])
