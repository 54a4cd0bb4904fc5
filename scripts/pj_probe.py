import sys, faulthandler
sys.path.insert(0, '.')
faulthandler.dump_traceback_later(50, exit=True)
import numpy as np
from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
sc, data, halos = make_chunk(20000, 8, seed=21)
d, h = to_numpy(data, halos)
cf = CategoryFilter(dmo=True)
for radii in [(30.0,), (10.0, 30.0, 300.0)]:
    props = chp.build_halo_prop_list(sc, cf, {}, {}, projected_vars=chp.projected_variations(radii))
    eng = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(eng, d)
    ha = dict({k: v.copy() for k, v in h.items()}, done=np.zeros(len(h["index"]), dtype=np.int32))
    res = []
    print("running", radii, flush=True)
    out = halo_tasks.process_halos(eng, props, sc, ha, res)
    print("done", radii, out[2], out[3], out[4]["status"], flush=True)
    print({k: v[4] for k, v in res[0].items() if "projx" in k and "30kpc" in k}, flush=True)
    eng.close()
