"""Halo-phase times of a workload with the classes alone and together (diagnostic): SOAP_ENGINE_ONLY_CLASS = 3 (small-halo
class) / 0 / 1 / 2, then everything."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = dict(bench.WORKLOADS[os.environ.get("WL", "c3")])
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
SMALL_ENV = os.environ.get("SOAP_ENGINE_SMALL")
configs = os.environ.get("CONFIGS", "only3,only0,only1,only2,all,nosmall").split(",")
for cfg in configs:
    os.environ.pop("SOAP_ENGINE_ONLY_CLASS", None)
    if SMALL_ENV is None:
        os.environ.pop("SOAP_ENGINE_SMALL", None)
    else:
        os.environ["SOAP_ENGINE_SMALL"] = SMALL_ENV   # (the caller's choice, except in the "nosmall" configuration)
    if cfg.startswith("only"):
        os.environ["SOAP_ENGINE_ONLY_CLASS"] = cfg[4:]
    if cfg == "nosmall":
        os.environ["SOAP_ENGINE_SMALL"] = "0"
    eng = halo_tasks.make_engine(sc, device=0)
    ms = []
    for rep in range(4):
        halo_tasks.stage_chunk(eng, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        try:
            out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
        except RuntimeError:
            pass   # (one class alone: the halos escalated to another class stay unfinished)
        ms.append([eng.kernel_ms(k) for k in ("halo", "halo_small", "halo_wave", "halo_wg256", "halo_wg1024", "halo_small512")])
    ms = np.array(ms[1:]).mean(axis=0)
    print(f"{cfg:8s} halo {ms[0]:6.2f}  small {ms[1]:6.2f} / {ms[5]:6.2f}  wave {ms[2]:6.2f}  wg256 {ms[3]:6.2f}  wg1024 {ms[4]:6.2f}   n_small {int(eng.kernel_ms('n_small128'))} + {int(eng.kernel_ms('n_small512'))} + {int(eng.kernel_ms('n_small1024'))} ({eng.kernel_ms('halo_small1024'):.2f} ms) esc {int(eng.kernel_ms('n_small_escalated'))} "
          f"n_wave {int(eng.kernel_ms('n_wave'))} n256 {int(eng.kernel_ms('n_wg256'))} n1024 {int(eng.kernel_ms('n_wg1024'))} passes {int(eng.kernel_ms('n_passes'))}", flush=True)
    eng.close()
