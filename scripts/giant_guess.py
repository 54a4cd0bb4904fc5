"""How good is k_predict's radius guess for the halos of the giants' pipeline (diagnostic)?  Per halo above the pipeline limit: estimate,
guessed radius of the first superset, catalogue search / read radius, radius the loop ended at, loop rounds.  WL as for bench.py (default c3u)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3u")]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=int(os.environ.get("SEED", "1234")), device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"),
                             topcells=bench.TOPCELLS, float32_masses=wl.get("float32_masses", False))
props = bench.build_props(sc, wl)[4]
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
os.environ["SOAP_ENGINE_DUMP_EST"] = "/tmp/est.bin"
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
raw = np.fromfile("/tmp/est.bin", dtype=np.uint8)
est = raw[:8 * H].view(np.int64); first = raw[8 * H:16 * H].view(np.float64)
stats = out[4]["stats"]
lim = int(os.environ.get("LIMIT", "262144"))
big = np.flatnonzero(est > lim)
big = big[np.argsort(-est[big])]
print("row est n_sel guess/search0 final/search0 read0/search0 final/guess rounds restarts")
for i in big:
    s0 = hn["search_radius"][i]
    print(f"{i:6d} {est[i]:8d} {stats[i, 1]:8d} {first[i] / s0:6.2f} {ha['search_radius'][i] / s0:6.2f} {hn['read_radius'][i] / s0:6.2f} "
          f"{(ha['search_radius'][i] / first[i]) if first[i] > 0 else float('nan'):6.2f} {stats[i, 0]:3d} {stats[i, 5]:2d}")
