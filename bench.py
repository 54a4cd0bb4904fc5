#!/usr/bin/env python
"""
bench.py -- headline benchmark of the halo-property engine (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2], the one the north-star target is
quoted on -- synthetic DMO chunk, 1e8 particles / 1e5 halos, BoundSubhalo + the 9 default SO
definitions + inclusive/exclusive 30/50/100 kpc spheres (16 calculations), fp64, one chunk per
GPU.  `--workload c2` selects configs[1] (1e7 / 1e4, SO_200_crit only).

A "step" is one pass of the hot path over the chunk with the raw particle arrays already
resident in HBM: mesh build + reorder into cell order (soap_engine_stage_particles) and
process_halos (soap_engine_run) -- SURVEY.md section 8(d).  Multi-GPU: weak scaling, one chunk per
rank, no data-path collective; the per-rank result tables are concatenated with ONE RCCL
all-gather (the catalogue gather that replaces combine_chunks' parallel sort), inside the
timed region.

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

WORKLOADS = {
    "c3": dict(n_part=100_000_000, n_halo=100_000, so="default", apertures=(30.0, 50.0, 100.0),
               name="synthetic DMO 1e8 particles / 1e5 halos, BoundSubhalo + 9 SO + incl/excl 30/50/100 kpc"),
    "c2": dict(n_part=10_000_000, n_halo=10_000, so="200_crit", apertures=(),
               name="synthetic DMO 1e7 particles / 1e4 halos, BoundSubhalo + SO_200_crit"),
    "c5": dict(n_part=25_000_000, n_halo=25_000, so="default", apertures=(30.0, 50.0, 100.0),
               projected=(10.0, 30.0, 50.0, 100.0), hydro=True,
               name="synthetic hydro (gas+DM+stars+BH) 2.5e7 particles / 2.5e4 halos per GPU (2e8 on 8 GPUs), "
                    "BoundSubhalo + 9 SO + incl/excl 30/50/100 kpc + projected 10/30/50/100 kpc"),
    "tiny": dict(n_part=400_000, n_halo=400, so="default", apertures=(30.0, 50.0, 100.0),
                 name="synthetic DMO 4e5 particles / 4e2 halos (plumbing check)"),
}  # fmt: skip


class _NamedPropertySet:
    """Parameter-file stand-in (core/parameter_file.py:85-154): the benchmark configurations name the SO /
    aperture property set of the north star (masses, centres, angular momenta, dispersion tensors, half-mass
    radii, v_max, concentration, flow rates); the inertia tensors (up to 20 passes per tensor over
    the search sphere) are switched off, for the engine and for the CPU baseline alike."""

    def get_property_filters(self, base_halo_type, names):
        if os.environ.get("BENCH_INERTIA") == "1":   # diagnostic: the same configurations with the tensors on
            return {n: "basic" for n in names}
        return {n: (False if "InertiaTensor" in n else "basic") for n in names}


def build_props(sc, wl):
    from soap_amd import compute_halo_properties as chp
    from soap_amd.core.category_filter import CategoryFilter

    cf = CategoryFilter(dmo=not wl.get("hydro", False))
    so_vars = None if wl["so"] == "default" else {"200_crit": {"value": 200.0, "type": "crit"}}
    ap_vars = chp.aperture_variations(wl["apertures"]) if wl["apertures"] else {}
    pj_vars = chp.projected_variations(wl["projected"]) if wl.get("projected") else {}
    return cf, so_vars, ap_vars, pj_vars, chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars,
                                                                   parameters=_NamedPropertySet())


def cpu_baseline(sc, data, halos, wl, budget_s=20.0):
    """Oracle (CPU restatement of the reference, one core) timed on a bounded sample of the
    SAME chunk: the halos whose read sphere lies inside a sub-cube, with the particles of that
    sub-cube copied to the host.  Only process_halos-equivalent work is timed."""
    import torch

    from soap_amd.core.category_filter import CategoryFilter
    from tests import helpers
    from oracle import halo_oracle as ho

    L = sc.boxsize
    n = sum(int(d["Coordinates"].shape[0]) for d in data.values())
    # about 3e6 particles: 10-20 s of oracle work on one core for the C3 / C5 property sets (bounded by budget_s)
    frac = min(1.0, (3.0e6 / n) ** (1.0 / 3.0))
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
    cf, so_vars, ap_vars, pj_vars, _ = build_props(sc, wl)
    u, props = helpers.oracle_prop_list(sc, cf, so_vars, ap_vars, pj_vars)
    ho.ENABLE_INERTIA = os.environ.get("BENCH_INERTIA") == "1"   # same property set as the engine run (_NamedPropertySet)
    npart = sum(d["Masses"].shape[0] for d in d_np.values())
    mesh = {}
    for pt, d in d_np.items():
        n_pt = int(data[pt]["Coordinates"].shape[0])
        mesh[pt] = ho.Mesh(d["Coordinates"], max(1, int(round(ho.mesh_resolution(n_pt) * frac))))
    A = len(props)
    td = ho.target_density_of(props, u)
    t0 = time.perf_counter()
    evals = 0
    nh = 0
    for i in range(len(h_np["index"])):
        ih = {k: v[i, ...].copy() for k, v in h_np.items()}
        st = {}
        r = ho.process_single_halo(mesh, d_np, props, u, ih, td if ih["is_central"] == 1 else None, st)
        if r is not None:
            evals += st["n_sel"] * A
            nh += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {
        "value": evals / dt, "unit": "evals/s", "cores": 1, "kind": "port",
        "halos_per_s": nh / dt,
        "sample": f"{nh} halos of the same chunk inside a {frac:.3f}L sub-cube ({npart} particles), "
                  f"{dt:.1f} s of NumPy oracle on 1 core; the true reference adds unyt overhead",
    }  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g

    g.build()
    from soap_amd.core import halo_tasks
    from soap_amd.synthetic import make_chunk

    wl = WORKLOADS[args.workload]
    t_gen = time.time()
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=args.seed + rank, device=f"cuda:{local_rank}",
                                 hydro=wl.get("hydro", False))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    cf, so_vars, ap_vars, pj_vars, props = build_props(sc, wl)
    A = len(props)
    halos_np = {k: v.cpu().numpy() for k, v in halos.items()}
    H = len(halos_np["index"])

    engine = halo_tasks.make_engine(sc, device=local_rank)
    ncols_total = None
    gathered = None

    # catalogue gather over xGMI: one all-gather of the (equal-sized) result tables per step.  Chunks are
    # independent, so the gather of step k runs (async, on RCCL's stream, double-buffered) while step k+1
    # stages and processes its chunk; every gather is complete before the timed region ends (sync()).
    pending = [None, None]
    bufs = [None, None]
    k_step = 0

    def step():
        nonlocal ncols_total, gathered, k_step
        halo_tasks.stage_chunk(engine, data)
        ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        raw = out[4]
        if world > 1:
            nc = engine.row_columns
            b = k_step % 2
            if pending[b] is not None:
                pending[b].wait()
            if bufs[b] is None:
                bufs[b] = (torch.empty(H * nc, dtype=torch.float64, device=f"cuda:{local_rank}"),
                           torch.empty(world * H * nc, dtype=torch.float64, device=f"cuda:{local_rank}"))
            engine.copy_table_to(bufs[b][0])
            pending[b] = dist.all_gather_into_tensor(bufs[b][1], bufs[b][0], async_op=True)
            gathered = bufs[b][1]
            k_step += 1
        return raw, out[2]

    def sync():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Catalogue radii in their steady state: halos whose calculations need a larger region than the catalogue's
    # read radius come back with new radii (halo_tasks.py:393-402) and are re-run, as ChunkTask does
    # (chunk_tasks.py:306-340).  Done once here, untimed, so that every timed step finishes every halo in one
    # pass (a handful of 1e5: the outer flow-rate shell of a low-density SO reaches past the read radius).
    ha0 = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.stage_chunk(engine, data)
    for _ in range(6):
        if halo_tasks.process_halos(engine, props, sc, ha0, None, unpack=False)[2] == 0:
            break
    n_radius_updates = int(np.sum(ha0["read_radius"] != halos_np["read_radius"]))
    halos_np["read_radius"] = ha0["read_radius"]
    halos_np["search_radius"] = ha0["search_radius"]

    for _ in range(args.warmup):
        step()
    ms_halo, ms_mesh, ms_reorder = [], [], []
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        raw, nleft = step()
        ms_halo.append(engine.kernel_ms("halo"))
        ms_mesh.append(engine.kernel_ms("mesh"))
        ms_reorder.append(engine.kernel_ms("reorder"))
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    stats = raw["stats"]
    ok = raw["status"] == 0
    n_sel = int(stats[ok, 1].sum())
    n_cand = int(stats[ok, 2].sum())
    evals = n_sel * A
    tot = torch.tensor([evals, int(ok.sum()), n_cand], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    evals_all, halos_all, ncand_all = (float(x) for x in tot.tolist())

    if rank == 0:
        ncols = engine.row_columns
        t_halo = float(np.mean(ms_halo)) * 1e-3
        b_alg = n_cand * B_PARTICLE + H * ncols * 8
        achieved = b_alg / t_halo / 1e9
        # HBM-side bytes of the halo kernels from the committed PMC passes (rocprofv3 cannot run inside the
        # bench): only quoted when the workload is the one that was profiled
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "round1", "final_c3_hbm_traffic.json")) as f:
                pj = json.load(f)
            if pj.get("workload") == wl["name"] and pj.get("seed") == args.seed and world == 1:
                traffic = pj["traffic_bytes_per_step_corrected"]
        except (OSError, ValueError, KeyError):
            traffic = None
        line = {
            "metric": "particle_aperture_evals_per_s",
            "value": evals_all * args.steps / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl["name"], "particles_per_gpu": wl["n_part"],
                       "halos_per_gpu": H, "calculations": A, "seed": args.seed,
                       "mesh_resolution": int(halo_tasks.mesh_resolution(wl["n_part"]))},
            "halos_per_s": halos_all * args.steps / dt,
            "halos_unfinished": int(nleft), "catalogue_radius_updates": n_radius_updates,
            "kernel_ms": {"halo": float(np.mean(ms_halo)), "mesh": float(np.mean(ms_mesh)),
                          "reorder": float(np.mean(ms_reorder)),
                          "halo_wave": engine.kernel_ms("halo_wave"), "halo_wg256": engine.kernel_ms("halo_wg256"),
                          "halo_wg1024": engine.kernel_ms("halo_wg1024")},
            "halos_per_class": [int(engine.kernel_ms("n_wave")), int(engine.kernel_ms("n_wg256")),
                                int(engine.kernel_ms("n_wg1024"))],
            "roofline": {
                "bound": "hbm", "kernel": "k_halo", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes": b_alg,
                "note": "sum_h n_cand(h)*68 B + H*P*8 B per step / HIP-event duration of the three concurrent k_halo "
                        "launches (g64/g256/g1024); traffic = 2*FETCH_SIZE + WRITE_SIZE of those launches from "
                        "profiles/round1/final_c3_hbm_traffic.json (separate --pmc passes, gfx950 read correction)",
            },
            "reorder_roofline": {
                "achieved": 2 * wl["n_part"] * B_PARTICLE / (float(np.mean(ms_reorder)) * 1e-3) / 1e9,
                "unit": "GB/s", "algorithmic_bytes": 2 * wl["n_part"] * B_PARTICLE,
            },
            "generation_s": t_gen,
            "phase_cycles": engine.phase_cycles(),
            "n_loop_mean": float(stats[ok, 0].mean()), "n_sel_max": int(stats[ok, 1].max()),
            "n_sel_sum": n_sel, "n_cand_sum": n_cand,
        }  # fmt: skip
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(sc, data, halos, wl)
            except Exception as e:  # the baseline must never hide the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
