"""Phase split (shader clocks of thread 0, SOAP_ENGINE_PROFILE=1) of the giant workload's halo kernels."""
import os, sys
os.environ.setdefault("SOAP_ENGINE_PROFILE", "1")   # n > 1: only the property stage of pipeline halos with n particles and more
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
from soap_amd import engine as eng
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "giant"]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
e = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(e, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.process_halos(e, props, sc, ha, None, unpack=False, resident_radius=np.inf)
pc = e.phase_cycles()
tot = pc.get("stage_b", 0) or 1
for k, v in sorted(pc.items(), key=lambda kv: -kv[1]):
    if v > 0:
        print("%-18s %.3g cycles  %.1f%% of stage_b" % (k, v, 100 * v / tot))
print("halo ms", e.kernel_ms("halo"), "B", e.kernel_ms("pipe_stage_b"))
