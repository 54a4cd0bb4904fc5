"""Radius of a particle that sits exactly on the halo centre: engine vs NumPy's wrap arithmetic (diagnostic)."""
import sys, dataclasses
sys.path.insert(0, ".")
import numpy as np
from soap_amd.units import SnapshotScalars
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd import compute_halo_properties as chp
from tests import helpers
L = 40.0
sc = dataclasses.replace(SnapshotScalars(), boxsize=L)
rng = np.random.default_rng(1)
n = 64
cs = rng.uniform(0, L, (n, 3))
cs[0] = [4.010538230515167, 3.78875346803561, 28.110248921115254]
cs[1:33, 0] = rng.uniform(0, 16, 32); cs[1:33, 1] = rng.uniform(0, 16, 32); cs[1:33, 2] = rng.uniform(0, 16, 32)
pos = np.concatenate([cs, rng.uniform(0, L, (2000, 3))])
N = len(pos)
grnr = np.full(N, -1, dtype=np.int32); grnr[:n] = np.arange(n) + 5
data = {1: {"Coordinates": pos, "Velocities": np.zeros((N, 3)), "Masses": np.full(N, 1.0), "GroupNr_bound": grnr,
            "FOFGroupIDs": np.full(N, 1, dtype=np.int64)}}
halos = {"index": (np.arange(n) + 5).astype(np.int64), "cofp": cs.copy(), "search_radius": np.full(n, 0.5), "read_radius": np.full(n, 3.0),
         "is_central": np.ones(n, dtype=np.int32), "nr_bound_part": np.ones(n, dtype=np.int64)}
cf = CategoryFilter(dmo=True)
props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, {})
eng = halo_tasks.make_engine(sc, device=0)
halo_tasks.stage_chunk(eng, data)
ha = dict({k: v.copy() for k, v in halos.items()}, done=np.zeros(n, dtype=np.int32))
res = []
halo_tasks.process_halos(eng, props, sc, ha, res)
off = cs - 0.5 * L
w = ((cs - off) % L) + off
rel = w - cs
r_np = np.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2 + rel[:, 2] ** 2)
got = np.array([r["BoundSubhalo/EncloseRadius"][4][0] for r in res])
done = np.flatnonzero(ha["done"] == 1)
bad = np.flatnonzero(got != r_np[done])
print("halos", len(done), "mismatching radii", len(bad))
for b in bad[:5]:
    print(got[b], r_np[done][b], rel[done][b])
