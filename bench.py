#!/usr/bin/env python
"""
bench.py -- headline benchmark of the halo-property engine (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2], the one the north-star target is
quoted on -- synthetic DMO box, 1e8 particles / 1e5 halos, BoundSubhalo + the 9 default SO
definitions + inclusive/exclusive 30/50/100 kpc spheres (16 calculations), fp64.
`--workload c2` selects configs[1] (1e7 / 1e4, SO_200_crit only), `c5` the hydro chunk.

One PASS over the box = what ChunkTask.__call__ does for a chunk whose particles are already
resident in HBM (SOAP/core/chunk_tasks.py:189-367): mesh build + reorder into cell order
(soap_engine_stage_particles) and process_halos (soap_engine_run), repeated for the halos that
come back with a larger read radius (halo_tasks.py:393-402) until none is left -- the retry passes
are inside the timed region.  A timed STEP is `--passes-per-step` (default 8) consecutive passes,
so that the driver's 20 steps keep the GPU busy for several seconds; `ms_per_pass` is the
round-1 "ms_per_step".

Multi-GPU (BASELINE.json configs[3]): STRONG scaling, the SAME seed-1234 box on every N.  The halos are cut
into N Peano-Hilbert chunks (soap_amd/core/domain_decomposition.py), rank g gets chunk g and every particle
of the top-level cells its halos' read spheres overlap (boundary cells replicated, no particle exchange:
soap_amd/core/chunk_tasks.py), read once before the timed region (the stand-in of the snapshot read); the
per-rank catalogues -- the columns of the result table that are stored, multiplied by their comoving -> stored-unit
factor and cast to the stored dtype on the device (`PackedLayout.pack`) -- are concatenated and put in
final-catalogue order by `gather_catalogue_packed` (one all-gather per dtype over RCCL/xGMI plus a device sort)
inside the timed region.  Until the driver's SCALE run
exists there is no hardware scaling curve; `replication` in the line says how many particles the chunks
hold together relative to the box.

metric = particle-aperture evaluations per second: E = sum_h n_sel(h) * A, n_sel = particles in
the halo's final search sphere, A = number of calculations.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
B_PARTICLE = 68  # x,y,z,vx,vy,vz,m fp64 + GroupNr_bound i32 + FOFGroupIDs i64
# top-level cells per dimension of the synthetic snapshot (soap_amd/synthetic.py; SWIFT boxes of this size are written with 32-64).
# A chunk reads whole cells half a cell beyond its halos' read spheres (mask_cells.py:30-34), so the grid sets how many
# particles the ranks of a multi-GPU run hold twice
TOPCELLS = int(os.environ.get("BENCH_TOPCELLS", "32"))
# how `--gpus N` cuts the Peano-Hilbert curve into ranks: "work" = equal estimated work (domain_decomposition.halo_work_weights),
# "count" = the reference's equal halo count (domain_decomposition.py:135-140)
RANK_CUT = os.environ.get("BENCH_RANK_CUT", "count")

WORKLOADS = {
    "c3": dict(n_part=100_000_000, n_halo=100_000, so="default", apertures=(30.0, 50.0, 100.0),
               name="synthetic DMO 1e8 particles / 1e5 halos, BoundSubhalo + 9 SO + incl/excl 30/50/100 kpc; "
                    "inertia tensors off; halo sizes capped at 3e5 particles"),
    "c2": dict(n_part=10_000_000, n_halo=10_000, so="200_crit", apertures=(),
               name="synthetic DMO 1e7 particles / 1e4 halos, BoundSubhalo + SO_200_crit; "
                    "inertia tensors off; halo sizes capped at 3e5 particles"),
    "c5": dict(n_part=25_000_000, n_halo=25_000, so="default", apertures=(30.0, 50.0, 100.0),
               projected=(10.0, 30.0, 50.0, 100.0), hydro=True, float32_masses=True,
               name="synthetic hydro (gas+DM+stars+BH) 2.5e7 particles / 2.5e4 halos (the per-GPU share of 2e8 on 8 "
                    "GPUs), masses float32-valued as SWIFT stores them, BoundSubhalo + 9 SO + incl/excl 30/50/100 kpc + "
                    "projected 10/30/50/100 kpc; inertia tensors off"),
    "giant": dict(n_part=20_000_000, n_halo=2_000, so="default", apertures=(30.0, 50.0, 100.0), max_halo=1_000_000,
                  name="synthetic DMO 2e7 particles / 2e3 halos with the halo size cap lifted to 1e6 particles, "
                       "C3 calculation list; inertia tensors off"),
    "c3u": dict(n_part=100_000_000, n_halo=100_000, so="default", apertures=(30.0, 50.0, 100.0), max_halo=1_000_000,
                name="synthetic DMO 1e8 particles / 1e5 halos, BoundSubhalo + 9 SO + incl/excl 30/50/100 kpc; "
                     "inertia tensors off; halo sizes capped at 1e6 particles (the `uncapped` entry of full_set on its own)"),
    "tiny": dict(n_part=400_000, n_halo=400, so="default", apertures=(30.0, 50.0, 100.0),
                 name="synthetic DMO 4e5 particles / 4e2 halos (plumbing check)"),
}  # fmt: skip


class _NamedPropertySet:
    """Parameter-file stand-in (core/parameter_file.py:85-154): the benchmark configurations name the SO /
    aperture property set of the north star (masses, centres, angular momenta, dispersion tensors, half-mass
    radii, v_max, concentration, flow rates); the inertia tensors (up to 20 passes per tensor over
    the search sphere) are switched off, for the engine and for the CPU baseline alike."""

    def __init__(self, inertia=False):
        self.inertia = inertia or os.environ.get("BENCH_INERTIA") == "1"   # (the environment switch: a diagnostic)

    def get_property_filters(self, base_halo_type, names):
        if self.inertia:   # the same configurations with the tensors on (the `full_set` block of the JSON line)
            return {n: "basic" for n in names}
        return {n: (False if "InertiaTensor" in n else "basic") for n in names}

    def strict_halo_copy(self):
        return False


def build_props(sc, wl, inertia=False):
    from soap_amd import compute_halo_properties as chp
    from soap_amd.core.category_filter import CategoryFilter

    cf = CategoryFilter(dmo=not wl.get("hydro", False))
    so_vars = None if wl["so"] == "default" else {"200_crit": {"value": 200.0, "type": "crit"}}
    ap_vars = chp.aperture_variations(wl["apertures"]) if wl["apertures"] else {}
    pj_vars = chp.projected_variations(wl["projected"]) if wl.get("projected") else {}
    return cf, so_vars, ap_vars, pj_vars, chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars,
                                                                   parameters=_NamedPropertySet(inertia))


# ---- CPU baseline: the oracle on the host cores ------------------------------------------------------------
def _cpu_worker(wid, payload_path, counter, budget_s, out):
    """One worker per core pulling halo numbers off a shared counter, biggest halos first
    (halo_tasks.py:352-357, chunk_tasks.py:118-120).  Started with `spawn`: a fresh interpreter that only ever
    imports NumPy / SciPy and the oracle -- nothing in it touches the GPU."""
    import pickle

    from oracle import halo_oracle as ho

    with open(payload_path, "rb") as f:
        g = pickle.load(f)
    ho.ENABLE_INERTIA = g["inertia"]
    n = len(g["h_np"]["index"])
    with counter.get_lock():   # rendezvous: the clock starts when every worker has loaded the sample
        counter.value += 1
    while counter.value < g["nworkers"]:
        time.sleep(0.005)
    t0 = time.perf_counter()
    evals = nh = 0
    while True:
        with counter.get_lock():
            i = counter.value - g["nworkers"]
            counter.value += 1
        # the workers go round the sample (biggest first each lap) until the budget is spent: many cores finish one
        # lap of a sample that fits the host in a second or two, too short a measurement
        if i >= n * g["max_laps"] or time.perf_counter() - t0 > budget_s:
            break
        i = i % n
        ih = {k: v[i, ...].copy() for k, v in g["h_np"].items()}
        st = {}
        r = ho.process_single_halo(g["mesh"], g["d_np"], g["props"], g["u"], ih,
                                   g["td"] if ih["is_central"] == 1 else None, st)  # fmt: skip
        if r is not None:
            evals += st["n_sel"] * g["A"]
            nh += 1
    out[wid] = (evals, nh, time.perf_counter() - t0)


def _cgroup_cpu_quota():
    """CPU quota of this process' cgroup in cores (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), or None when there is
    none: the affinity mask of a container usually shows the whole host, the quota is what the scheduler really grants."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            return max(1, int(round(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0 and per > 0:
            return max(1, int(round(q / per)))
    except (OSError, ValueError):
        pass
    return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def subcube_sample(sc, data, halos, target_particles=3.0e6):
    """A bounded sample of a box for the CPU oracle: the halos whose read sphere lies inside the sub-cube [0, f L]^3 that holds
    about `target_particles` particles, and the particles of that sub-cube on the host (in their staged order, so that ties of
    the (r, index) sort fall the same way).  Returns (rows of the halo table, particle arrays, halo arrays, f) or None."""
    import torch

    L = sc.boxsize
    n = sum(int(d["Coordinates"].shape[0]) for d in data.values())
    frac = min(1.0, (target_particles / n) ** (1.0 / 3.0))
    side = L * frac
    rr = halos["read_radius"]
    c = halos["cofp"]
    inside = ((c - rr[:, None] >= 0.0) & (c + rr[:, None] <= side)).all(dim=1)
    rows = torch.nonzero(inside).flatten()
    if rows.numel() == 0:
        return None
    d_np = {}
    for pt, d in data.items():
        psel = (d["Coordinates"] <= side).all(dim=1)
        d_np[pt] = {k: v[psel].cpu().numpy() for k, v in d.items()}
    h_np = {k: v[rows].cpu().numpy() for k, v in halos.items()}
    return rows.cpu().numpy(), d_np, h_np, frac


def cpu_baseline(sc, data, halos, wl, budget_s=12.0):
    """Oracle (CPU restatement of the reference) timed on a bounded sample of the SAME box: the halos whose
    read sphere lies inside a sub-cube, with the particles of that sub-cube copied to the host.  Only
    process_halos-equivalent work is timed: first on one core, then with one worker per core of this
    process' CPU share pulling halos from a shared counter."""
    import multiprocessing as mp

    import torch

    from oracle import halo_oracle as ho
    from tests import helpers

    sample = subcube_sample(sc, data, halos)
    if sample is None:
        return None
    rows, d_np, h_np, frac = sample
    order = np.argsort(h_np["nr_bound_part"])[::-1]
    h_np = {k: v[order] for k, v in h_np.items()}
    cf, so_vars, ap_vars, pj_vars, _ = build_props(sc, wl)
    u, props = helpers.oracle_prop_list(sc, cf, so_vars, ap_vars, pj_vars)
    ho.ENABLE_INERTIA = os.environ.get("BENCH_INERTIA") == "1"   # same property set as the engine run
    npart = sum(d["Masses"].shape[0] for d in d_np.values())
    mesh = {}
    for pt, d in d_np.items():
        n_pt = int(data[pt]["Coordinates"].shape[0])
        mesh[pt] = ho.Mesh(d["Coordinates"], max(1, int(round(ho.mesh_resolution(n_pt) * frac))))
    import pickle
    import tempfile

    payload = dict(d_np=d_np, h_np=h_np, props=props, u=u, mesh=mesh, A=len(props), td=ho.target_density_of(props, u),
                   inertia=os.environ.get("BENCH_INERTIA") == "1", nworkers=1, max_laps=1)
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ctx = mp.get_context("spawn")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    tmpdir = tempfile.mkdtemp(prefix="soap_cpu_baseline_")
    path = os.path.join(tmpdir, "sample.pkl")

    def run(nworkers, budget):
        payload["nworkers"] = nworkers
        payload["max_laps"] = 1 if nworkers == 1 else 64
        with open(path, "wb") as f:
            pickle.dump(payload, f, protocol=4)
        counter = ctx.Value("l", 0)
        with ctx.Manager() as mgr:
            out = mgr.dict()
            ps = [ctx.Process(target=_cpu_worker, args=(w, path, counter, budget, out)) for w in range(nworkers)]
            for p in ps:
                p.start()
            for p in ps:
                p.join()
            res = dict(out)
        if len(res) != nworkers:
            raise RuntimeError(f"{nworkers - len(res)} CPU baseline workers died")
        dt = max(v[2] for v in res.values())
        ev = sum(v[0] for v in res.values())
        nh = sum(v[1] for v in res.values())
        return ev / dt, nh / dt, nh, dt

    try:
        v1, h1, n1, t1 = run(1, budget_s * 0.5)
        # one worker per core of this process' share of the host: the cgroup's CPU quota when there is one (a one-GPU box of the
        # pool gets 16 cores of a 256-thread host, and its affinity mask does not say so), else 16; BENCH_CPU_CORES overrides
        quota = _cgroup_cpu_quota()
        ncore = max(1, min(share, int(os.environ.get("BENCH_CPU_CORES", str(quota if quota else 16)))))
        vN, hN, nN, tN = run(ncore, budget_s) if ncore > 1 else (v1, h1, n1, t1)
    finally:
        import shutil

        shutil.rmtree(tmpdir, ignore_errors=True)
    return {
        "value": vN, "unit": "evals/s", "cores": ncore, "kind": "port", "halos_per_s": hN,
        "one_core": {"value": v1, "halos_per_s": h1},
        "host": {"cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "cpu_share_of_this_process": share, "cgroup_cpu_quota": quota},
        "sample": f"{len(h_np['index'])} halos of the same box inside a {frac:.3f}L sub-cube ({npart} particles), biggest "
                  f"first off a shared counter, lap after lap until the time budget is spent: {nN} halos in {tN:.1f} s on "
                  f"{ncore} worker processes ({n1} in {t1:.1f} s on one); NumPy oracle, the true reference adds unyt overhead",
    }  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--passes-per-step", type=int, default=8)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-set", action="store_true", help="skip the full_set block (tensors on, halo size cap lifted)")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal knob (one-GPU boxes): BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo, to exercise
    # the multi-rank code path without RCCL; the real launch is one rank per GPU over RCCL
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g

    g.build()
    from soap_amd import layout
    from soap_amd.catalogue import gather_catalogue_packed, spatial_sort_order_device
    from soap_amd.core.result_set import PackedLayout
    from soap_amd.core import chunk_tasks, halo_tasks
    from soap_amd.synthetic import make_chunk

    wl = WORKLOADS[args.workload]
    t_gen = time.time()
    # every rank builds the SAME box (same seed) and keeps its chunk's region of it
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=args.seed, device=dev, hydro=wl.get("hydro", False),
                                 max_halo=wl.get("max_halo"), topcells=TOPCELLS, float32_masses=wl.get("float32_masses", False))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    cf, so_vars, ap_vars, pj_vars, props = build_props(sc, wl)

    def measure(wl, sc, data, halos, props, n_steps, n_warmup):
        """n_warmup untimed and n_steps timed steps of passes_per_step passes over one box with one calculation list; returns the
        locals the JSON line is made of (the engine of the measurement still open: the caller reads its clocks and closes it)"""
        A = len(props)
        halos_all = {k: v.cpu().numpy() for k, v in halos.items()}
        snapshot = chunk_tasks.SnapshotCells(data, sc.boxsize, (TOPCELLS,) * 3)
        n_box = snapshot.nr_particles()
        if world > 1:
            halos_np = chunk_tasks.split_halos(halos_all, sc.boxsize, world, weights="work" if RANK_CUT == "work" else None)[rank]
            region, _, mask0 = chunk_tasks.read_chunk(snapshot, halos_np, wrap=True, return_mask=True)
            # how far the cells this rank read are complete around each of its halos: read-radius updates that stay
            # inside need no re-read (soap_engine_set_resident_radius)
            resident0 = chunk_tasks.resident_radius(mask0, snapshot.cell_size, halos_np["cofp"], halos_np["read_radius"])
            torch.cuda.synchronize()
        else:
            halos_np = halos_all
            region = data
            resident0 = np.full(len(halos_np["index"]), np.inf)   # the whole periodic box is resident
        H = len(halos_np["index"])
        n_region = sum(int(d["Coordinates"].shape[0]) for d in region.values())
        cell_size = snapshot.cell_size
        d_cofp = torch.as_tensor(halos_np["cofp"], device=dev)
        d_index = torch.as_tensor(halos_np["index"], device=dev)

        engine = halo_tasks.make_engine(sc, device=local_rank)
        state = dict(nleft=0, npass=0, ncols=None, raw=None, catalogue=None, table=None)
        packed = PackedLayout(props, engine.desc_columns)
        ms_halo, ms_mesh, ms_reorder = [], [], []

        def one_pass(record):
            """ChunkTask loop for the resident region: stage, process, and again for the halos that need a larger
            read radius (their cells are read again, chunk_tasks.py:189-367)."""
            ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
            cur, resident = region, resident0
            stats = status = order = None
            table = state["table"]
            t_h = 0.0
            for it in range(8):
                halo_tasks.stage_chunk(engine, cur)
                if it == 0 and world == 1 and os.environ.get("BENCH_ORDER_EARLY", "1") != "0":
                    # the (cell, catalogue index) order of the final catalogue only needs the halo centres: queued on torch's stream now,
                    # it runs while the host prepares the halo table (the staging call has returned with its placement pass in flight)
                    order = spatial_sort_order_device(d_cofp, d_index, cell_size, (TOPCELLS,) * 3)
                out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=resident)
                if it == 0 and record:
                    ms_mesh.append(engine.kernel_ms("mesh"))
                    ms_reorder.append(engine.kernel_ms("reorder"))
                raw = out[4]
                t_h += engine.kernel_ms("halo")
                rows = raw["rows"]
                okm = raw["status"] == 0
                if stats is None and rows.size == H and okm.all():
                    stats, status = raw["stats"], raw["status"]   # every halo finished in the first call: the engine's arrays as they are
                else:
                    if stats is None:
                        stats = np.zeros((H, layout.STATS_COLUMNS), dtype=np.int64)
                        status = np.full(H, -99, dtype=np.int32)
                    stats[rows[okm]] = raw["stats"][okm]
                    status[rows] = raw["status"]
                direct = it == 0 and rows.size == H and okm.all()   # every halo finished in the first call: the stored catalogue is packed straight from the engine's table
                if direct:
                    state["npass"] += 1
                    break
                if table is None:
                    state["ncols"] = engine.row_columns
                    table = state["table"] = torch.zeros((H, state["ncols"]), dtype=torch.float64, device=dev)
                # finished rows into this pass' catalogue table (device to device)
                if rows.size == H:
                    engine.copy_table_to(table.view(-1))
                else:
                    part = torch.empty((rows.size, state["ncols"]), dtype=torch.float64, device=dev)
                    engine.copy_table_to(part.view(-1))
                    sel = torch.as_tensor(rows[okm], device=dev)
                    table[sel] = part[torch.as_tensor(np.flatnonzero(okm), device=dev)]
                state["npass"] += 1
                if out[2] == 0:
                    break
                cur, _, mk = chunk_tasks.read_chunk(snapshot, ha, wrap=world > 1, return_mask=True)
                resident = chunk_tasks.resident_radius(mk, snapshot.cell_size, ha["cofp"], ha["read_radius"])
            if record:
                ms_halo.append(t_h)
            state["nleft"] = int(np.sum(ha["done"] == 0))
            if state["nleft"]:   # (the chunk loop of the reference runs until no halo is left: a pass that leaves some is not a measurement)
                raise RuntimeError(f"{state['nleft']} halos unfinished after 8 read-radius iterations")
            state["raw"] = dict(stats=stats, status=status, iterations=it + 1,
                                radius_updates=int(np.sum(ha["read_radius"] != halos_np["read_radius"])))
            # final catalogue: concatenation over the ranks in (cell, catalogue index) order
            if direct and world == 1:
                # one rank: the engine writes the stored columns in final-catalogue order (soap_engine_pack_catalogue)
                o = order if order is not None else spatial_sort_order_device(d_cofp, d_index, cell_size, (TOPCELLS,) * 3)
                narrow, wide = packed.pack_engine(engine, dev, order=o)
                state["catalogue"] = (narrow, wide, d_cofp[o], d_index[o])
            else:
                narrow, wide = packed.pack_engine(engine, dev) if direct else packed.pack(table)
                state["catalogue"] = gather_catalogue_packed(narrow, wide, d_cofp, d_index, cell_size, (TOPCELLS,) * 3)

        def sync():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        one_pass(False)   # engine set-up: memory pools, slabs and torch's allocator reach their working size (never timed)
        for _ in range(n_warmup):   # a warm-up STEP is what a timed step is: passes_per_step passes
            for _ in range(args.passes_per_step):
                one_pass(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            for _ in range(args.passes_per_step):
                one_pass(True)
        sync()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        npasses = n_steps * args.passes_per_step

        raw = state["raw"]
        stats = raw["stats"]
        ok = raw["status"] == 0
        n_sel = int(stats[ok, 1].sum())
        n_cand = int(stats[ok, 2].sum())
        evals = n_sel * A
        tot = torch.tensor([evals, int(ok.sum()), n_cand, n_region, state["nleft"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        evals_all, halos_all_ok, ncand_all, nregion_all, nleft_all = (float(x) for x in tot.tolist())
        cat_table = state["catalogue"][0]
        assert cat_table.shape[0] == wl["n_halo"], (cat_table.shape, wl["n_halo"])

        return dict(locals())

    M = measure(wl, sc, data, halos, props, args.steps, args.warmup)
    (engine, packed, state, raw, stats, ok, n_sel, n_cand, A, H, dt, npasses, n_region, n_box, evals_all, halos_all_ok, nregion_all,
     nleft_all, ms_halo, ms_mesh, ms_reorder) = (M[k] for k in (
        "engine", "packed", "state", "raw", "stats", "ok", "n_sel", "n_cand", "A", "H", "dt", "npasses", "n_region", "n_box", "evals_all",
        "halos_all_ok", "nregion_all", "nleft_all", "ms_halo", "ms_mesh", "ms_reorder"))

    def halo_counts(engine):
        return {
            "halos_by_path": {"pipeline": int(engine.kernel_ms("n_pipeline")), "single_kernel": int(engine.kernel_ms("n_single")),
                              "teams": int(engine.kernel_ms("n_team"))},
            # halos whose first sorted superset, class and place in the launch order follow k_predict's guess of where their radius loop ends
            "halos_with_a_radius_guess": int(engine.kernel_ms("n_predicted")),
            "single_kernel_passes": int(engine.kernel_ms("n_passes")),   # more than 1: a halo outgrew its slab and was done again
            "halos_per_class": [int(engine.kernel_ms("n_wave")), int(engine.kernel_ms("n_wg256")),
                                int(engine.kernel_ms("n_wg1024"))],
            # the small-halo class (one wavefront per halo, slab in LDS): halos that went through its 128- / 512-entry kernels, and
            # how many of them were handed on to the wavefront class (counted there too)
            "halos_small_class": {"slab128": int(engine.kernel_ms("n_small128")), "slab512": int(engine.kernel_ms("n_small512")),
                                  "slab1024": int(engine.kernel_ms("n_small1024")),
                                  "handed_on": int(engine.kernel_ms("n_small_escalated"))},
        }  # fmt: skip

    def variant_block(name, wl_v, sc_v, data_v, halos_v, inertia):
        """one entry of `full_set`: the same measurement (same passes, same catalogue gather) with another calculation list or box"""
        props_v = build_props(sc_v, wl_v, inertia=inertia)[4]
        V = measure(wl_v, sc_v, data_v, halos_v, props_v, 1, 0)
        eng = V["engine"]
        t_halo = float(np.mean(V["ms_halo"])) * 1e-3
        b_alg = V["n_cand"] * B_PARTICLE + V["H"] * eng.row_columns * 8
        blk = {
            "workload": name, "calculations": V["A"], "row_columns": int(eng.row_columns), "passes": V["npasses"],
            "value": V["evals_all"] * V["npasses"] / V["dt"], "unit": "evals/s",
            "ms_per_pass": V["dt"] / V["npasses"] * 1e3,
            "halos_unfinished": int(V["nleft_all"]),
            "kernel_ms": {"halo": float(np.mean(V["ms_halo"])), "mesh": float(np.mean(V["ms_mesh"])), "reorder": float(np.mean(V["ms_reorder"]))},
            "roofline": {"bound": "hbm", "achieved": b_alg / t_halo / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b_alg / t_halo / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b_alg, "traffic": None},
            "n_sel_max": int(V["stats"][V["ok"], 1].max()), "n_sel_sum": V["n_sel"], "n_cand_sum": V["n_cand"],
            **halo_counts(eng),
        }  # fmt: skip
        eng.close()
        return blk

    if rank == 0:
        ncols = engine.row_columns
        t_halo = float(np.mean(ms_halo)) * 1e-3
        b_alg = n_cand * B_PARTICLE + H * ncols * 8
        achieved = b_alg / t_halo / 1e9
        # HBM-side bytes of the halo kernels from the committed PMC passes (rocprofv3 cannot run inside the
        # bench): only quoted when the workload is the one that was profiled
        traffic = None
        tnote = "no PMC profile of this workload committed"
        for prof in ("round5", "round4", "round3", "round2", "round1"):
            try:
                with open(os.path.join(ROOT, "profiles", prof, "final_c3_hbm_traffic.json")) as f:
                    pj = json.load(f)
                if pj.get("workload_key", "c3") == args.workload and pj.get("seed") == args.seed and world == 1:
                    traffic = pj["traffic_bytes_per_step_corrected"]
                    tnote = f"2*FETCH_SIZE + WRITE_SIZE of the halo kernels from profiles/{prof}/final_c3_hbm_traffic.json (separate --pmc passes, gfx950 read correction)"
                    break
            except (OSError, ValueError, KeyError):
                continue
        # the placement pass: coordinates read (24 B), cell id + rank read back (8 B), coordinates written in cell order (24 B),
        # velocities / masses / ids read (44 B) and the 80-byte payload record written (the original index travels in it)
        b_reorder = (wl["n_part"] if world == 1 else n_region) * (24 + 8 + 24 + 44 + 80)
        line = {
            "metric": "particle_aperture_evals_per_s",
            "value": evals_all * npasses / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl["name"], "workload_key": args.workload, "particles": wl["n_part"],
                       "halos": wl["n_halo"], "calculations": A, "seed": args.seed,
                       "passes_per_step": args.passes_per_step,
                       "chunks": world, "top_level_cells": TOPCELLS, "rank_cut": RANK_CUT if world > 1 else None, "mesh_resolution": int(halo_tasks.mesh_resolution(n_region))},
            "ms_per_pass": dt / npasses * 1e3,
            "halos_per_s": halos_all_ok * npasses / dt,
            "halos_unfinished": int(nleft_all),
            "process_halos_calls_per_pass": raw["iterations"], "catalogue_radius_updates": raw["radius_updates"],
            "replication": nregion_all / n_box,
            "catalogue_bytes_per_halo": {"stored": packed.bytes_per_halo(), "fp64_table": 8 * packed.table_columns},
            "kernel_ms": {"halo": float(np.mean(ms_halo)), "mesh": float(np.mean(ms_mesh)),
                          "reorder": float(np.mean(ms_reorder)),
                          "halo_small128": engine.kernel_ms("halo_small"), "halo_small512": engine.kernel_ms("halo_small512"),
                          "halo_small1024": engine.kernel_ms("halo_small1024"),
                          "halo_wave": engine.kernel_ms("halo_wave"), "halo_wg256": engine.kernel_ms("halo_wg256"),
                          "halo_wg1024": engine.kernel_ms("halo_wg1024"),
                          "pipe_build": engine.kernel_ms("pipe_build"), "pipe_stage_a": engine.kernel_ms("pipe_stage_a"),
                          "pipe_stage_b": engine.kernel_ms("pipe_stage_b"),
                          "pipe_build_by_class": [engine.kernel_ms("pipe_build_" + c) for c in ("wave", "wg256", "wg1024")],
                          "pipe_props_by_class": [engine.kernel_ms("pipe_props_" + c) for c in ("wave", "wg256", "wg1024")]},
            **halo_counts(engine),
            **({"phase_cycles": engine.phase_cycles()} if os.environ.get("SOAP_ENGINE_PROFILE") == "1" else {}),   # diagnostics
            "roofline": {
                "bound": "hbm", "kernel": "k_halo + k_small (the concurrent launches of the halo phase)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes": b_alg,
                "note": "rank 0, per pass: sum_h n_cand(h)*68 B + H*P*8 B / HIP-event duration of the halo kernels of all "
                        "process_halos calls of the pass (concurrent class launches, events on the engine's streams); traffic: " + tnote,
            },
            "reorder_roofline": {
                "achieved": b_reorder / (float(np.mean(ms_reorder)) * 1e-3) / 1e9,
                "unit": "GB/s", "algorithmic_bytes": b_reorder,
                "note": "bytes the placement pass moves per particle: coordinates read (24 B) and written in cell order (24 B), "
                        "cell id + rank read back (8 B), velocities / masses / ids read (44 B) and the "
                        "80-byte payload record written",
            },
            "generation_s": t_gen,
            "n_loop_mean": float(stats[ok, 0].mean()), "n_sel_max": int(stats[ok, 1].max()),
            "n_sel_sum": n_sel, "n_cand_sum": n_cand,
        }  # fmt: skip
        engine.close()
        engine = None
        M.clear()
        if world == 1 and not args.no_full_set and args.workload == "c3":
            # BASELINE.json's "full" property set and catalogue shape, measured in this process after the headline configuration:
            # the inertia tensors on (property_calculation/inertia_tensors.py:19-132, up to 20 passes per tensor), the halo size cap of
            # the synthetic catalogue lifted to 1e6 particles (the giants go through the multi-workgroup pipeline), and both
            try:
                full = {"note": "same process, same pass as the headline line (stage, halo kernels, catalogue gather), 1 untimed "
                                "set-up pass + 1 step of passes_per_step passes each"}
                full["inertia_on"] = variant_block(wl["name"].replace("inertia tensors off", "inertia tensors ON"), wl, sc, data, halos, True)
                wl_u = dict(wl, max_halo=1_000_000, name=wl["name"].replace("capped at 3e5", "capped at 1e6"))
                sc_u, data_u, halos_u = make_chunk(wl_u["n_part"], wl_u["n_halo"], seed=args.seed, device=dev, max_halo=wl_u["max_halo"],
                                                           topcells=TOPCELLS)
                full["uncapped"] = variant_block(wl_u["name"], wl_u, sc_u, data_u, halos_u, False)
                full["uncapped_inertia_on"] = variant_block(wl_u["name"].replace("inertia tensors off", "inertia tensors ON"),
                                                            wl_u, sc_u, data_u, halos_u, True)
                del data_u, halos_u
                line["full_set"] = full
            except Exception as e:   # never hide the headline number
                line["full_set"] = {"error": repr(e)}
            # the other one-GPU configurations of BASELINE.json under the same clock, each with its own roofline block: configs[1]
            # (c2: 1e7 / 1e4, BoundSubhalo + SO_200_crit) and the per-GPU share of configs[4] (c5: hydro, projected apertures)
            other = {"note": "same process and measurement as full_set (1 untimed set-up pass + 1 step of passes_per_step passes), boxes of "
                             "their own: `python bench.py --workload c2|c5` gives the same numbers as a headline line"}
            for key in ("c2", "c5"):
                try:
                    wl_o = WORKLOADS[key]
                    sc_o, data_o, halos_o = make_chunk(wl_o["n_part"], wl_o["n_halo"], seed=args.seed, device=dev, hydro=wl_o.get("hydro", False),
                                                       max_halo=wl_o.get("max_halo"), topcells=TOPCELLS, float32_masses=wl_o.get("float32_masses", False))
                    other[key] = variant_block(wl_o["name"], wl_o, sc_o, data_o, halos_o, False)
                    del data_o, halos_o
                except Exception as e:
                    other[key] = {"error": repr(e)}
            line["other_configs"] = other
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(sc, data, halos, wl)
            except Exception as e:  # the baseline must never hide the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if engine is not None:
        engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
