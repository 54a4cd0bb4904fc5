"""What each rank of `bench.py --gpus N` would do, one rank after the other on ONE GPU: its share of the halos of
the C3 box (Peano-Hilbert chunks), the particles of the cells those halos can reach, and the time of a pass over it.
The N-GPU pass time is the maximum over the ranks (plus the catalogue all-gather)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.core import chunk_tasks, halo_tasks
from soap_amd.synthetic import make_chunk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wlname = sys.argv[2] if len(sys.argv) > 2 else "c3"
wl = bench.WORKLOADS[wlname]
dev = "cuda:0"
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device=dev, hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS)
cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
halos_all = {k: v.cpu().numpy() for k, v in halos.items()}
snapshot = chunk_tasks.SnapshotCells(data, sc.boxsize, (bench.TOPCELLS,) * 3)
n_box = snapshot.nr_particles()
engine = halo_tasks.make_engine(sc, device=0)
chunks = chunk_tasks.split_halos(halos_all, sc.boxsize, N, weights="work" if bench.RANK_CUT == "work" else None) if N > 1 else [halos_all]
worst = 0.0
tot_region = 0
only = [int(x) for x in os.environ["RANKS"].split(",")] if os.environ.get("RANKS") else None
for r, hn in enumerate(chunks):
    if only is not None and r not in only:
        continue
    if N > 1:
        region, _, mask0 = chunk_tasks.read_chunk(snapshot, hn, wrap=True, return_mask=True)
        resident0 = chunk_tasks.resident_radius(mask0, snapshot.cell_size, hn["cofp"], hn["read_radius"])
    else:
        region, resident0 = data, np.full(len(hn["index"]), np.inf)
    H = len(hn["index"])
    n_region = sum(int(d["Coordinates"].shape[0]) for d in region.values())
    tot_region += n_region
    def one_pass():
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        cur, resident = region, resident0
        calls = 0
        for it in range(8):
            halo_tasks.stage_chunk(engine, cur)
            out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=resident)
            calls += 1
            if out[2] == 0:
                break
            cur, _, mk = chunk_tasks.read_chunk(snapshot, ha, wrap=True, return_mask=True)
            resident = chunk_tasks.resident_radius(mk, snapshot.cell_size, ha["cofp"], ha["read_radius"])
        return calls
    one_pass(); one_pass()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nrep = 6
    each = []
    for _ in range(nrep):
        t1 = time.perf_counter()
        calls = one_pass()
        torch.cuda.synchronize(); each.append(round(1e3 * (time.perf_counter() - t1), 1))
    ms = float(np.median(each))   # (the first pass of a rank now and then pays for fresh allocations)
    print("   passes:", each, "host run_setup/pre/classify:", [round(engine.kernel_ms(k), 2) for k in ("run_setup", "run_pre", "run_classify")])
    worst = max(worst, ms)
    km = {k: round(engine.kernel_ms(k), 2) for k in ("mesh", "reorder", "halo", "halo_wave", "halo_wg256", "halo_wg1024", "pipe_build", "pipe_stage_a", "pipe_stage_b", "n_pipeline", "n_single", "run_total")}
    print(f"rank {r}/{N}: halos {H} particles {n_region} ({n_region / n_box:.3f} of the box) calls/pass {calls} ms/pass {ms:.2f} last call: {km}", flush=True)
    del region
print(f"N={N} {wlname}: slowest rank {worst:.2f} ms per pass; replication {tot_region / n_box:.3f}")
