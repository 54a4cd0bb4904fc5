import sys, os
sys.path.insert(0, '.')
os.environ["SOAP_ENGINE_DEBUG_STOP"] = "15"
import numpy as np
from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
sc, data, halos = make_chunk(3000, 2, seed=3)
d, hh = to_numpy(data, halos)
cf = CategoryFilter(dmo=True)
props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, {})
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, d)
calcs = [hp.calc_desc(props, i) for i, hp in enumerate(props)]
eng.stage_halos(hh["cofp"], hh["search_radius"], hh["read_radius"], hh["index"], hh["is_central"], hh["nr_bound_part"])
eng.run(calcs, halo_tasks.target_density_of(props, sc))
print({k: hex(v) for k, v in eng.phase_cycles().items()})
