"""The longest per-halo chains of ONE rank's share of the C3 box (stats column 3: shader clocks of the halo's thread group): RANK=4 python scripts/rank_chain_top.py 8"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import chunk_tasks, halo_tasks
from soap_amd.synthetic import make_chunk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(os.environ.get("RANK", "4"))
wl = bench.WORKLOADS["c3"]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", topcells=bench.TOPCELLS)
props = bench.build_props(sc, wl)[4]
halos_all = {k: v.cpu().numpy() for k, v in halos.items()}
snapshot = chunk_tasks.SnapshotCells(data, sc.boxsize, (bench.TOPCELLS,) * 3)
hn = chunk_tasks.split_halos(halos_all, sc.boxsize, N)[r]
region, _, mask0 = chunk_tasks.read_chunk(snapshot, hn, wrap=True, return_mask=True)
resident0 = chunk_tasks.resident_radius(mask0, snapshot.cell_size, hn["cofp"], hn["read_radius"])
H = len(hn["index"])
engine = halo_tasks.make_engine(sc, device=0)
dump = "/tmp/est.bin"
os.environ["SOAP_ENGINE_DUMP_EST"] = dump
for rep in range(2):
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.stage_chunk(engine, region)
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=resident0)
raw = np.fromfile(dump, dtype=np.int64)
est, first = raw[:H], raw[H:2 * H].view(np.float64)
st = out[4]["stats"]
clk = st[:, 3].astype(np.float64)
print("halo kernels %.2f ms (wg256 %.2f, pipeline build %.2f)" % (engine.kernel_ms("halo"), engine.kernel_ms("halo_wg256"), engine.kernel_ms("pipe_build")))
print("   row  clocks/1e5    n_sel    n_cand      est  guess_R  nr_bound  loops attempts")
for i in np.argsort(-clk)[:16]:
    print(f"{i:6d} {clk[i] / 1e5:9.2f} {st[i,1]:9d} {st[i,2]:9d} {est[i]:8d} {first[i]:8.3f} {hn['nr_bound_part'][i]:9d} {st[i,4]:6d} {st[i,5]:4d}")
