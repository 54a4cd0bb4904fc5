"""Which path is right where the small-halo class and the general path disagree? (diagnostic: both against the oracle)"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
from tests import helpers
from tests.test_small_halo_gpu import _NoInertia
from oracle import halo_oracle as ho

sc, data, halos = make_chunk(250000, 400, seed=71)
data_np, halos_np = to_numpy(data, halos)
cf = CategoryFilter(dmo=True)
ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
props = chp.build_halo_prop_list(sc, cf, None, ap_vars, parameters=_NoInertia())
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, data_np)
res = {}
for mode in ("0", "3"):
    os.environ["SOAP_ENGINE_SMALL"] = mode
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    r = []
    out = halo_tasks.process_halos(eng, props, sc, ha, r)
    res[mode] = (r, eng.fetch(table=True)["table"], out[4]["stats"].copy())
ho.ENABLE_INERTIA = False
ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars)
a, b = res["3"][1], res["0"][1]
scale = np.maximum(np.abs(a), np.abs(b)).max(axis=0) + 1e-300
rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-12 * scale)
rel[a == b] = 0
bad = np.flatnonzero(rel.max(axis=1) > 1e-9)
print("bad rows", bad)
for row in bad[:6]:
    cols = np.flatnonzero(rel[row] > 1e-9)
    print("row", row, "n_sel", res["0"][2][row, 1], "nr_bound", halos_np["nr_bound_part"][row], "cols", cols)
    o = ref[row]
    for key in o:
        if "HalfMassRadiusDM" not in key and not key.endswith("/Ndm"):
            continue
        grp, k2 = key.rsplit("/", 1)
        print("   ", key, "oracle", o[key].raw, "general", helpers.engine_value(res["0"][0][row], grp, k2), "small", helpers.engine_value(res["3"][0][row], grp, k2))
