"""The longest halos of ONE class of a workload (diagnostic): the class runs alone (SOAP_ENGINE_ONLY_CLASS=k: 0 wavefront, 1 256 threads,
2 1024 threads, 3 small-halo kernels), the other halos stay unprocessed; prints the class clock and the rows with the most shader clocks.
   WL=c3u python scripts/class_top.py 1"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

k = sys.argv[1] if len(sys.argv) > 1 else "1"
key = os.environ.get("WL", "c3")
wl = bench.WORKLOADS[key]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=int(os.environ.get("SEED", "1234")), device="cuda:0", hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS, float32_masses=os.environ.get("F32_MASSES") == "1" or wl.get("float32_masses", False))
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
hn = {kk: v.cpu().numpy() for kk, v in halos.items()}
H = len(hn["index"])
os.environ["SOAP_ENGINE_ONLY_CLASS"] = k
os.environ["SOAP_ENGINE_DUMP_EST"] = "/tmp/soap_est.bin"
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(2):
    halo_tasks.stage_chunk(eng, data)
    ha = dict({kk: v.copy() for kk, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    try:
        out = halo_tasks.process_halos(eng, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    except RuntimeError as e:
        print("run:", e); break
st = out[4]["stats"]
clk = st[:, 3]
print("class", k, "halo kernels %.2f ms" % eng.kernel_ms("halo"), {n: round(eng.kernel_ms(n), 2) for n in ("halo_wave", "halo_wg256", "halo_wg1024", "halo_small", "halo_small512")}, "rows with clocks", int((clk > 0).sum()))
raw = np.fromfile("/tmp/soap_est.bin", dtype=np.int64)
est, first = raw[:H], raw[H:].view(np.float64)
for i in np.argsort(-clk)[:16]:
    print(f"row {i:6d} clk {clk[i] / 2.4e6:7.2f} ms(2.4 GHz) nb {hn['nr_bound_part'][i]:7d} n_sel {st[i, 1]:7d} n_cand {st[i, 2]:7d} loops {st[i, 0]}/{st[i, 4]} attempts {st[i, 5]} Rs {hn['search_radius'][i]:.3f} -> {ha['search_radius'][i]:.3f} central {hn['is_central'][i]} estimate {est[i]} guess {first[i]:.3f} read {hn['read_radius'][i]:.3f}")
