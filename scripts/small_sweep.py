"""Halo-phase time of a workload under a list of environment settings (diagnostic): CASES="A=1,B=2;C=3;..." """
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = dict(bench.WORKLOADS[os.environ.get("WL", "c3")])
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=int(os.environ.get("SEED", "1234")), device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
for case in os.environ.get("CASES", "").split(";"):
    env = dict(kv.split("=") for kv in case.split(",") if kv)
    for k, v in env.items():
        os.environ[k] = v
    eng = halo_tasks.make_engine(sc, device=0)
    ms = []
    for rep in range(4):
        halo_tasks.stage_chunk(eng, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
        ms.append([eng.kernel_ms(k) for k in ("halo", "halo_small", "halo_small512", "halo_wave", "halo_wg256", "halo_wg1024", "pipe_build", "pipe_stage_b")])
    m = np.array(ms[1:]).mean(axis=0)
    print(f"{case:60s} halo {m[0]:6.2f} small {m[1]:5.2f}/{m[2]:5.2f} wave {m[3]:5.2f} wg256 {m[4]:5.2f} wg1024 {m[5]:5.2f} pipe {m[6]:5.2f}/{m[7]:5.2f} "
          f"n: {int(eng.kernel_ms('n_small128'))}+{int(eng.kernel_ms('n_small512'))} esc {int(eng.kernel_ms('n_small_escalated'))} / {int(eng.kernel_ms('n_wave'))} / {int(eng.kernel_ms('n_wg256'))} / {int(eng.kernel_ms('n_wg1024'))} pipe {int(eng.kernel_ms('n_pipeline'))} passes {int(eng.kernel_ms('n_passes'))} ok {int((out[4]['status'] == 0).sum())}", flush=True)
    eng.close()
    for k in env:
        os.environ.pop(k, None)
