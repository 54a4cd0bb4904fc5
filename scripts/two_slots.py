"""Two chunk slots on one GPU: two engine handles, one host thread each, alternating passes over the same box (diagnostic for the
bench's `slots`).  LOCK=run (only one halo phase at a time; default) | none | ph (whole process_halos); SLOTS=1 is the serial loop.
WL as for bench.py."""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soap_amd.catalogue import spatial_sort_order_device
from soap_amd.core.result_set import PackedLayout
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk

wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
SLOTS = int(os.environ.get("SLOTS", "2"))
LOCK = os.environ.get("LOCK", "run")
NPASS = int(os.environ.get("NPASS", "16"))
dev = torch.device("cuda:0")
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device=dev, hydro=wl.get("hydro", False), max_halo=wl.get("max_halo"),
                             topcells=bench.TOPCELLS, float32_masses=wl.get("float32_masses", False))
props = bench.build_props(sc, wl)[4]
hn = {k: v.cpu().numpy() for k, v in halos.items()}
H = len(hn["index"])
d_cofp = torch.as_tensor(hn["cofp"], device=dev); d_index = torch.as_tensor(hn["index"], device=dev)
cell = np.array([sc.boxsize / bench.TOPCELLS] * 3)
engines = [halo_tasks.make_engine(sc, device=0) for _ in range(SLOTS)]
packed = PackedLayout(props, engines[0].desc_columns)
lock = threading.Lock()
for e in engines:
    if LOCK == "run":
        real = e.run
        def locked(*a, _real=real, **k):
            with lock:
                return _real(*a, **k)
        e.run = locked

halo_ms = [[] for _ in engines]
def one_pass(s, record):
    e = engines[s]
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.stage_chunk(e, data)
    o = spatial_sort_order_device(d_cofp, d_index, cell, (bench.TOPCELLS,) * 3)
    if LOCK == "ph":
        with lock:
            out = halo_tasks.process_halos(e, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    else:
        out = halo_tasks.process_halos(e, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    assert out[2] == 0
    if record:
        halo_ms[s].append(e.kernel_ms("halo"))
    narrow, wide = packed.pack_engine(e, dev, order=o)
    return narrow, wide

def worker(s, n, record):
    torch.cuda.set_device(0)
    for _ in range(n):
        one_pass(s, record)

def run_all(n, record):
    th = [threading.Thread(target=worker, args=(s, n // SLOTS, record)) for s in range(SLOTS)]
    for t in th: t.start()
    for t in th: t.join()

run_all(2 * SLOTS, False)
for rep in range(3):
    for h in halo_ms: h.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_all(NPASS, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"slots {SLOTS} lock {LOCK}: {dt / NPASS * 1e3:.2f} ms per pass, halo window {np.mean([np.mean(h) for h in halo_ms]):.2f} ms", flush=True)
