"""
Oracle parity AT FULL SIZE, on samples: the engine processes every halo of BASELINE.json's configurations (C3: 1e8 particles /
1e5 halos / 16 calculations; the C5 share: 2.5e7 particles of four types / 2.5e4 halos / 24 calculations; the `giant` box with
the halo size cap lifted to 1e6 particles) and the rows of a sample are compared with the oracle, halo by halo, value by value:
counts and statuses exactly, values to 1e-10 with the absolute floors of tests/test_parity_gpu.py (reference rule:
SOAP/core/halo_tasks.py:23-273).  The samples: the halos whose read sphere lies inside the sub-cube `bench.cpu_baseline` times the
oracle on (its particles are the oracle's whole data set, in staged order), and for the giants the largest halos, each with the
particles of its own read sphere.  What the sample does not cover stays held to the size-independent invariants of
test_parity_gpu.py::test_full_size_invariants_*.
"""

import time

import numpy as np
import pytest

from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
from tests import helpers
from tests.test_parity_gpu import compare_halo

pytestmark = pytest.mark.gpu


def _engine_pass(sc, data, halos, props):
    """every halo of the box through the engine, first attempt only (what one process_halos call of the reference's chunk loop does)"""
    hn = {k: v.cpu().numpy() for k, v in halos.items()}
    H = len(hn["index"])
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data)
    ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
    got = engine.fetch(table=True)
    counts = {k: int(engine.kernel_ms(k)) for k in ("n_small", "n_pipeline", "n_single", "n_wave", "n_wg256", "n_wg1024")}
    coloff = np.concatenate([[0], np.cumsum([engine.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)])])
    engine.close()
    return hn, ha, out[4], got, coloff, counts


def _compare_rows(sc, props, coloff, got, ha, rows, ref, stats_ref, ha_ref, no_inertia=True, dmo=True):
    """rows: rows of the engine's table the oracle's halos 0..n-1 are; ref: oracle results of the finished ones, in sample order"""
    st = got["status"][rows]
    assert np.array_equal(ha["done"][rows], ha_ref["done"]), "finished / needs-a-larger-read-radius differs"
    assert np.array_equal((st == 0).astype(np.int32), ha_ref["done"])
    assert np.array_equal(ha["read_radius"][rows], ha_ref["read_radius"])
    assert np.array_equal(ha["search_radius"][rows], ha_ref["search_radius"])
    done = np.flatnonzero(ha_ref["done"] == 1)
    assert len(done) == len(ref) == len(stats_ref)
    nchecked = 0
    for k, j in enumerate(done):
        hr = halo_tasks.unpack_row(props, coloff, got["table"][rows[j]])
        o = ref[k]
        if no_inertia:
            o = {kk: v for kk, v in o.items() if "InertiaTensor" not in kk}
        nchecked += compare_halo(o, hr, sc, dmo=dmo)
        s = got["stats"][rows[j]]
        # (radius rounds and sphere content; the candidates count mesh cells, and the oracle's mesh covers the sub-cube only)
        assert (s[0], s[1]) == (stats_ref[k]["n_loop"], stats_ref[k]["n_sel"]), (j, s[:3], stats_ref[k])
    return len(done), nchecked


def _subcube_parity(workload, min_halos, max_halos=None, inertia=False):
    import torch

    import bench
    from oracle import halo_oracle as ho

    t0 = time.time()
    wl = bench.WORKLOADS[workload]
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=wl.get("hydro", False),
                                 max_halo=wl.get("max_halo"), topcells=bench.TOPCELLS)  # fmt: skip
    cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl, inertia=inertia)
    hn, ha, raw, got, coloff, counts = _engine_pass(sc, data, halos, props)
    print(f"[sample {workload}] engine pass over {len(hn['index'])} halos done: {time.time() - t0:.1f} s, paths {counts}", flush=True)
    rows, d_np, h_np, frac = bench.subcube_sample(sc, data, halos)
    del data
    torch.cuda.empty_cache()
    if max_halos is not None and len(rows) > max_halos:   # an even cut through the sizes, the largest halo included
        order = np.argsort(-h_np["nr_bound_part"], kind="stable")
        keep = np.sort(order[np.unique(np.linspace(0, len(order) - 1, max_halos).astype(np.int64))])
        rows, h_np = rows[keep], {k: v[keep] for k, v in h_np.items()}
    assert len(rows) >= min_halos, (len(rows), min_halos)
    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = inertia   # the benchmark configurations have the tensors off (bench._NamedPropertySet); full_set has them on
    try:
        res = {pt: max(1, int(round(ho.mesh_resolution(int(wl["n_part"])) * frac))) for pt in d_np}
        ref, stats_ref, ha_ref = helpers.run_oracle(sc, d_np, h_np, cf, so_vars, ap_vars, resolution=max(res.values()),
                                                    projected_vars=pj_vars)  # fmt: skip
    finally:
        ho.ENABLE_INERTIA = saved
    print(f"[sample {workload}] oracle over {len(rows)} halos of the {frac:.3f} L sub-cube done: {time.time() - t0:.1f} s", flush=True)
    ref_by_pos = [ref[j] for j in np.flatnonzero(ha_ref["done"] == 1)]
    ndone, nchecked = _compare_rows(sc, props, coloff, got, ha, rows, ref_by_pos, stats_ref, ha_ref, no_inertia=not inertia, dmo=not wl.get("hydro", False))
    print(f"[sample {workload}] {ndone} halos, {nchecked} values compared: {time.time() - t0:.1f} s", flush=True)
    assert ndone >= 0.95 * len(rows)
    return counts


def test_c3_sample_matches_the_oracle(built_library):
    """C3 at full size (seed 1234, the box of the driver's bench line): >= 2000 halos of the cpu_baseline sub-cube, all 16 calculations"""
    counts = _subcube_parity("c3", min_halos=2000)
    assert counts["n_small"] > 50000   # the sample's small halos went through the small-halo class, as in the bench line


def test_c3_sample_with_the_inertia_tensors_matches_the_oracle(built_library):
    """the `full_set.inertia_on` configuration of the bench line: C3 with the 8 x 6 tensor columns of the bound subhalo and of every
    SO switched on -- grouped sweeps in the thread-group classes, the lane-per-tensor pass of the small-halo kernels -- against the
    oracle's iterative tensors on a cut through the sizes of the sub-cube sample"""
    counts = _subcube_parity("c3", min_halos=600, max_halos=700, inertia=True)
    assert counts["n_small"] > 50000


def test_c5_share_sample_matches_the_oracle(built_library):
    """the hydro configuration (one GPU's share of C5): gas, dark matter, stars and black holes, projected apertures included"""
    _subcube_parity("c5", min_halos=500, max_halos=1200)


@pytest.mark.parametrize("workload", ["giant", "c3u"])
def test_giant_workload_largest_halos_match_the_oracle(built_library, workload):
    """`--workload giant` (2e7 particles) and `c3u` (the C3 box, `full_set.uncapped` of the bench line), halo size cap 1e6 particles:
    the largest halos -- they go through the multi-workgroup pipeline or the 1024-thread class -- each against the oracle run on the
    particles of its own read sphere"""
    import torch

    import bench
    from oracle import halo_oracle as ho

    t0 = time.time()
    wl = bench.WORKLOADS[workload]
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", max_halo=wl["max_halo"], topcells=bench.TOPCELLS)
    cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
    hn, ha, raw, got, coloff, counts = _engine_pass(sc, data, halos, props)
    print(f"[sample {workload}] engine pass done: {time.time() - t0:.1f} s, paths {counts}", flush=True)
    order = np.argsort(-hn["nr_bound_part"], kind="stable")
    rows = np.sort(order[:10])
    assert hn["nr_bound_part"][rows].max() >= 400000
    L = sc.boxsize
    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    nchecked = ndone = 0
    try:
        u, oprops = helpers.oracle_prop_list(sc, cf, so_vars, ap_vars, pj_vars)
        td = ho.target_density_of(oprops, u)
        for row in rows:
            c = halos["cofp"][row]
            rr = float(halos["read_radius"][row])
            d_np = {}
            for pt, d in data.items():
                dx = d["Coordinates"] - c
                dx = dx - L * torch.round(dx / L)
                psel = (dx.abs() <= rr * 1.0001).all(dim=1)   # the cube around the read sphere, periodic
                d_np[pt] = {k: v[psel].cpu().numpy() for k, v in d.items()}
            mesh = {pt: ho.Mesh(d["Coordinates"], 16) for pt, d in d_np.items()}
            ih = {k: v[row, ...].copy() for k, v in hn.items()}
            st = {}
            r = ho.process_single_halo(mesh, d_np, oprops, u, ih, td if ih["is_central"] == 1 else None, st)
            assert (r is not None) == (got["status"][row] == 0), (row, got["status"][row])
            if r is None:
                continue
            hr = halo_tasks.unpack_row(props, coloff, got["table"][row])
            o = {kk: v for kk, v in r.items() if "InertiaTensor" not in kk and kk != "_counts"}
            nchecked += compare_halo(o, hr, sc)
            s = got["stats"][row]
            assert (s[0], s[1]) == (st["n_loop"], st["n_sel"]), (row, s[:3], st)   # (n_cand counts mesh cells: another mesh here)
            ndone += 1
            print(f"[sample {workload}] halo row {row}: {hn['nr_bound_part'][row]} bound, {st['n_sel']} in its sphere: {time.time() - t0:.1f} s", flush=True)
    finally:
        ho.ENABLE_INERTIA = saved
    assert ndone >= 8 and nchecked > 0
    assert counts["n_pipeline"] + counts["n_wg1024"] > 0
