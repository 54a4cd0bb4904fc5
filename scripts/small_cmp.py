"""Where do two runs of the test chunk differ? (diagnostic)  MODES=1,0"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
from tests.test_small_halo_gpu import _NoInertia

sc, data, halos = make_chunk(250000, 400, seed=71)
data_np, halos_np = to_numpy(data, halos)
cf = CategoryFilter(dmo=True)
props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 50.0, 100.0)), parameters=_NoInertia())
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, data_np)
res = {}
for mode in os.environ.get("MODES", "1,0").split(","):
    os.environ["SOAP_ENGINE_SMALL"] = mode
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False)
    res[mode] = (eng.fetch(table=True)["table"], out[4]["stats"].copy(), out[4]["status"].copy())
    print("mode", mode, "n_small", eng.kernel_ms("n_small128"), eng.kernel_ms("n_small512"), "esc", eng.kernel_ms("n_small_escalated"), "passes", eng.kernel_ms("n_passes"))
(a, sa, fa), (b, sb, fb) = res.values()
scale = np.maximum(np.abs(a), np.abs(b)).max(axis=0) + 1e-300
rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-12 * scale)
rel[a == b] = 0
bad = np.flatnonzero(rel.max(axis=1) > 1e-10)
print("bad rows", bad)
for row in bad[:10]:
    cols = np.flatnonzero(rel[row] > 1e-10)
    print("row", row, "n_sel", sa[row, 1], sb[row, 1], "nr_bound", halos_np["nr_bound_part"][row], "status", fa[row], fb[row], "ncols", len(cols), cols[:10], a[row, cols[:4]], b[row, cols[:4]])
