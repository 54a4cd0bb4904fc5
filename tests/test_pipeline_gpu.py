"""
The three-stage pipeline (build -> radius loop with one lane per halo -> property passes) against
the single-kernel path and against the oracle.  Same sorted slabs, same property passes: the two paths must give
the same table bit for bit, the same number of radius iterations and the same search spheres.
"""

import os

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
from tests import helpers
from tests.test_parity_gpu import compare_halo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _thread_group_classes_only(monkeypatch):
    """These tests compare the pipeline with the single-kernel path of the SAME thread-group classes bit for bit; the
    small-halo class (another order of the same sums, tests/test_small_halo_gpu.py) stays out of them."""
    monkeypatch.setenv("SOAP_ENGINE_SMALL", "0")


def engine_columns(engine, props, col):
    """name of the calculation a table column belongs to, and the column's offset inside its block (diagnostics)"""
    off = 0
    for i, p in enumerate(props):
        w = engine.desc_columns(p.calc_desc(props, i))
        if col < off + w:
            return (p.name, col - off)
        off += w
    return ("?", col)


class _NoInertia:
    """parameter-file stand-in: everything on except the inertia tensors (their iterative passes keep a run on the
    single-kernel path)"""

    def get_property_filters(self, base_halo_type, names):
        return {n: (False if "InertiaTensor" in n else "basic") for n in names}

    def strict_halo_copy(self):
        return False


def _run(engine, props, sc, halos_np, pipeline, resident=None):
    # 0: single-kernel path; 2: every halo through the pipeline (1 / unset: the default mix, small halos through the
    # pipeline beside the large ones on the single-kernel path)
    os.environ["SOAP_ENGINE_PIPELINE"] = str(int(pipeline))
    try:
        ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
        results = []
        out = halo_tasks.process_halos(engine, props, sc, ha, results, resident_radius=resident)
        table = engine.fetch(table=True)["table"]
        return ha, results, out[4], table, (engine.kernel_ms("n_pipeline"), engine.kernel_ms("n_single"))
    finally:
        os.environ.pop("SOAP_ENGINE_PIPELINE", None)


@pytest.mark.parametrize("hydro,seed,read_factor,nb_small,giant_sort",
                         [(False, 41, 8.0, None, None), (True, 42, 8.0, None, None), (False, 43, 1.5, None, None),
                          (False, 44, 8.0, 600, None), (True, 45, 1.5, 2000, None), (False, 46, 1.5, 600, 3000), (True, 47, 8.0, 2000, 5000)])
def test_pipeline_equals_single_kernel_path_and_oracle(built_library, monkeypatch, hydro, seed, read_factor, nb_small, giant_sort):
    if nb_small is not None:
        # most halos count as LARGE: their slabs are built by several workgroups each (k_g* kernels) in the pipeline,
        # by one 256-thread group on the single-kernel path
        monkeypatch.setenv("SOAP_ENGINE_NB_SMALL", str(nb_small))
    if giant_sort is not None:
        # ... and the larger of those are split into uniform radial bins instead of sample-sort buckets (the giants' sort)
        monkeypatch.setenv("SOAP_ENGINE_GIANT_SORT", str(giant_sort))
    # the bit-for-bit comparisons run without the tile sums of the property stage (k_bpartials: another order of the same
    # additions for the large halos); they are switched on for a run of their own below
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    sc, data, halos = make_chunk(250000, 180, seed=seed, hydro=hydro, satellite_fraction=0.1, read_radius_factor=read_factor)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=not hydro)
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    # (projected apertures also in one of the runs without gas / stars: its large halos must not be shared by two groups)
    pj_vars = chp.projected_variations((30.0, 100.0)) if (hydro or seed == 44) else {}
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars, projected_vars=pj_vars, parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha1, res1, raw1, t1, (np1, ns1) = _run(engine, props, sc, halos_np, pipeline=False)
    ha2, res2, raw2, t2, (np2, ns2) = _run(engine, props, sc, halos_np, pipeline=2)
    assert np1 == 0 and np2 > 0.8 * len(halos_np["index"]), (np1, np2, ns2)
    n_gbuild = engine.kernel_ms("n_gbuild")
    # the mix (small halos through the pipeline beside the large ones on the single-kernel path) gives the same again
    ha5, _, raw5, t5, (np5, ns5) = _run(engine, props, sc, halos_np, pipeline=1)
    assert (np5 > 0 or nb_small is not None) and np.array_equal(t5, t2) and np.array_equal(raw5["stats"][:, :3], raw2["stats"][:, :3])
    assert np.array_equal(ha5["read_radius"], ha2["read_radius"]) and np.array_equal(raw5["status"], raw2["status"])
    if nb_small is not None:
        assert n_gbuild > 0.3 * len(halos_np["index"])
    # giants mode (the default): halos above a limit through the pipeline, the others on the single-kernel path
    monkeypatch.setenv("SOAP_ENGINE_NB_GIANT", "4000")
    ha6, _, raw6, t6, (np6, ns6) = _run(engine, props, sc, halos_np, pipeline=3)
    monkeypatch.delenv("SOAP_ENGINE_NB_GIANT")
    assert np6 > 0 and np.array_equal(raw6["stats"][:, :3], raw2["stats"][:, :3])
    scale6 = np.maximum(np.abs(t2).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(t6 - t2) <= 1e-11 * np.abs(t2) + 1e-13 * scale6)   # the giants' sums run in 1024-thread groups here
    assert np.array_equal(ha6["read_radius"], ha2["read_radius"]) and np.array_equal(raw6["status"], raw2["status"])
    # the property stage of the large halos fed with tile sums computed by all CUs: same selections, sums to rounding
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "1")
    ha7, _, raw7, t7, (np7, ns7) = _run(engine, props, sc, halos_np, pipeline=2)
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    assert np7 == np2 and np.array_equal(raw7["stats"][:, :3], raw2["stats"][:, :3]) and np.array_equal(raw7["status"], raw2["status"])
    assert np.array_equal(ha7["read_radius"], ha2["read_radius"])
    assert np.all(np.abs(t7 - t2) <= 1e-11 * np.abs(t2) + 1e-13 * scale6)
    assert np.array_equal(raw1["status"], raw2["status"])
    assert np.array_equal(ha1["done"], ha2["done"])
    assert np.array_equal(ha1["read_radius"], ha2["read_radius"])
    assert np.array_equal(ha1["search_radius"], ha2["search_radius"])
    assert np.array_equal(raw1["stats"][:, :3], raw2["stats"][:, :3])   # n_loop, n_sel, n_cand
    assert np.array_equal(t1, t2)
    if read_factor < 2.0:
        assert np.any(raw2["status"] == 1)   # this chunk has halos that need a larger read radius
        # with the box declared resident both paths finish them in the same call, again identically
        ha3, _, raw3, t3, _ = _run(engine, props, sc, halos_np, pipeline=False, resident=np.inf)
        ha4, _, raw4, t4, (np4, ns4) = _run(engine, props, sc, halos_np, pipeline=2, resident=np.inf)
        assert np.all(ha3["done"] == 1) and np.all(ha4["done"] == 1)
        assert np.array_equal(ha3["read_radius"], ha4["read_radius"]) and np.array_equal(ha3["search_radius"], ha4["search_radius"])
        assert np.array_equal(raw3["stats"][:, :2], raw4["stats"][:, :2])
        scale = np.maximum(np.abs(t3).max(axis=0, keepdims=True), 1e-300)
        assert np.all(np.abs(t3 - t4) <= 1e-11 * np.abs(t3) + 1e-13 * scale)   # a rebuilt halo may change thread-group class
    engine.close()
    # and the oracle
    from oracle import halo_oracle as ho

    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    try:
        ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars, projected_vars=pj_vars)
    finally:
        ho.ENABLE_INERTIA = saved
    assert np.array_equal(ha2["done"], ha_ref["done"])
    done_rows = np.flatnonzero(ha2["done"] == 1)
    assert len(res2) == len(done_rows) == len(ref)
    for j, row in enumerate(done_rows):
        o = {k: v for k, v in ref[row].items() if "InertiaTensor" not in k}
        hr = res2[j]
        compare_halo(o, hr, sc) if not hydro else None
    st = raw2["stats"][raw2["status"] == 0]
    assert np.array_equal(st[:, 0], [s["n_loop"] for s in stats])
    assert np.array_equal(st[:, 1], [s["n_sel"] for s in stats])


@pytest.mark.parametrize("hydro,seed,n_part,n_halo", [(False, 51, 500000, 14), (True, 52, 500000, 14), (False, 54, 2000000, 24)])
def test_tile_sums_feed_the_property_stage_of_large_halos(built_library, monkeypatch, hydro, seed, n_part, n_halo):
    """Halos of tens of thousands of particles through the pipeline: the moment sums of whole 4096-entry tiles of their
    search spheres come from k_bpartials (all CUs) instead of the one or two groups of the property stage.  Same
    selections as without them, values to rounding, and the oracle's values.  The third case has a halo of 10^5
    particles: more than two tiles inside 0.3 R_SO, so the mass-flow sums of the inner spheres come from the tile sums too."""
    sc, data, halos = make_chunk(n_part, n_halo, seed=seed, hydro=hydro, satellite_fraction=0.1, read_radius_factor=8.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=not hydro)
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    pj_vars = chp.projected_variations((30.0, 100.0)) if hydro else {}
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars, projected_vars=pj_vars, parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    monkeypatch.setenv("SOAP_ENGINE_NB_SMALL", "2000")
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    ha0, res0, raw0, t0, (np0, ns0) = _run(engine, props, sc, halos_np, pipeline=2)
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "1")
    ha1, res1, raw1, t1, (np1, ns1) = _run(engine, props, sc, halos_np, pipeline=2)
    colname = lambda c: engine_columns(engine, props, c)  # noqa: E731
    assert np1 == np0 > 0 and raw1["stats"][:, 1].max() >= (5 * 4096 if n_part < 1000000 else 20 * 4096), raw1["stats"][:, 1].max()
    assert np.array_equal(raw1["stats"][:, :3], raw0["stats"][:, :3]) and np.array_equal(raw1["status"], raw0["status"])
    assert not np.array_equal(t1, t0)   # the sums did take the other route
    scale = np.maximum(np.abs(t0).max(axis=0, keepdims=True), 1e-300)
    excess = np.abs(t1 - t0) / (1e-11 * np.abs(t0) + 1e-11 * scale)   # (off-diagonal dispersion terms are differences of large sums)
    worst = np.unravel_index(np.argmax(excess), excess.shape)
    assert excess.max() <= 1.0, (worst, t0[worst], t1[worst], scale[0, worst[1]], colname(worst[1]))
    engine.close()
    from oracle import halo_oracle as ho

    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    try:
        ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars, projected_vars=pj_vars)
    finally:
        ho.ENABLE_INERTIA = saved
    assert np.array_equal(ha1["done"], ha_ref["done"])
    done_rows = np.flatnonzero(ha1["done"] == 1)
    assert len(res1) == len(done_rows) == len(ref)
    if not hydro:
        for j, row in enumerate(done_rows):
            compare_halo({k: v for k, v in ref[row].items() if "InertiaTensor" not in k}, res1[j], sc)
    st = raw1["stats"][raw1["status"] == 0]
    assert np.array_equal(st[:, 1], [s["n_sel"] for s in stats])


def test_oversized_sort_buckets_are_ranked(built_library, monkeypatch):
    """The tile-parallel build splits a large halo into buckets by splitters from a strided sample of the selection;
    a bucket that outgrows its LDS block is sorted by ranks instead of sending the halo to the single-group build.
    Forced here (four buckets per halo): same table as the single-kernel path, bit for bit, nobody falls back."""
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    monkeypatch.setenv("SOAP_ENGINE_NB_SMALL", "600")
    sc, data, halos = make_chunk(250000, 60, seed=53, satellite_fraction=0.1, read_radius_factor=8.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)), parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha1, _, raw1, t1, (np1, ns1) = _run(engine, props, sc, halos_np, pipeline=False)
    monkeypatch.setenv("SOAP_ENGINE_G_COARSE", "1")
    ha2, _, raw2, t2, (np2, ns2) = _run(engine, props, sc, halos_np, pipeline=2)
    n_gbuild = engine.kernel_ms("n_gbuild")
    engine.close()
    assert raw1["stats"][:, 1].max() > 4 * 2048 * 2   # buckets of more than two LDS blocks
    assert np1 == 0 and ns2 == 0 and np2 == len(halos_np["index"]) and n_gbuild > 20, (np1, np2, ns2, n_gbuild)
    assert np.array_equal(raw1["status"], raw2["status"]) and np.array_equal(raw1["stats"][:, :3], raw2["stats"][:, :3])
    assert np.array_equal(t1, t2)


def test_more_than_32_so_thresholds_through_the_tile_parallel_build(built_library, monkeypatch):
    """MAXCALC is 64: a run may carry more than 32 SO density definitions, and the tile kernel that tabulates the first
    entry below each threshold (k_gpost) keeps one bit per threshold and thread -- 64 of them.  With 40 thresholds the
    large halos built by several workgroups must give the table of the single-kernel path."""
    monkeypatch.setenv("SOAP_ENGINE_NB_SMALL", "600")
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    sc, data, halos = make_chunk(250000, 120, seed=48, read_radius_factor=8.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {f"{v:.0f}_crit": {"value": float(v), "type": "crit"} for v in np.linspace(60.0, 3180.0, 40).round()}
    props = chp.build_halo_prop_list(sc, cf, so_vars, {}, parameters=_NoInertia())
    assert sum(p.kind == 1 for p in props) == 40
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha1, _, raw1, t1, _ = _run(engine, props, sc, halos_np, pipeline=False)
    ha2, _, raw2, t2, (np2, _) = _run(engine, props, sc, halos_np, pipeline=2)
    assert np2 > 0.8 * len(halos_np["index"]) and engine.kernel_ms("n_gbuild") > 0.3 * len(halos_np["index"])
    assert np.array_equal(raw1["status"], raw2["status"]) and np.array_equal(raw1["stats"][:, :3], raw2["stats"][:, :3])
    assert np.array_equal(t1, t2)
    # the SO radii of the last thresholds (bits 32 .. 39 of the mask) are there and differ from those 32 places earlier
    cols = [engine.desc_columns(p.calc_desc(props, i)) for i, p in enumerate(props)]
    off = np.concatenate([[0], np.cumsum(cols)])
    r_hi, r_lo = t2[:, off[1 + 39]], t2[:, off[1 + 7]]
    ok = raw2["status"] == 0
    assert np.all(r_hi[ok] > 0.0) and np.all(r_hi[ok] < r_lo[ok])
    engine.close()


@pytest.mark.parametrize("hydro,seed,read_factor,nb_team,groups,giant_sort",
                         [(False, 51, 8.0, 600, 24, None), (True, 52, 1.5, 2000, 7, None), (False, 53, 1.5, 600, 16, 3000), (False, 54, 8.0, 600, 1, None)])
def test_teams_equal_the_single_kernel_path(built_library, monkeypatch, hydro, seed, read_factor, nb_team, groups, giant_sort):
    """SOAP_ENGINE_PIPELINE=4: the halos beyond nb_team candidates are done by the workgroups of ONE persistent launch
    (k_team: the phases of the tile-parallel build over all of them, the radius loops with lanes = halos, rebuilds further out,
    tile sums, the property passes, a grid barrier between the phases) beside the single-kernel path of the others.  Same
    sorted slabs, same property passes: the table of the single-kernel path bit for bit (tile sums off), to rounding with them;
    whatever the number of workgroups (1: every phase of every halo by one group)."""
    monkeypatch.setenv("SOAP_ENGINE_NB_TEAM", str(nb_team))
    monkeypatch.setenv("SOAP_ENGINE_TEAMS", str(groups))
    monkeypatch.setenv("SOAP_ENGINE_NB_SMALL", str(nb_team // 2))   # (the 256-thread class on the single-kernel side for the same halos)
    if giant_sort is not None:
        monkeypatch.setenv("SOAP_ENGINE_GIANT_SORT", str(giant_sort))
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "0")
    sc, data, halos = make_chunk(250000, 180, seed=seed, hydro=hydro, satellite_fraction=0.1, read_radius_factor=read_factor)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=not hydro)
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    pj_vars = chp.projected_variations((30.0, 100.0)) if hydro else {}
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars, projected_vars=pj_vars, parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    if engine.kernel_ms("experiments") != 1.0:
        engine.close()
        pytest.skip("k_team was measured slower than the launches (DESIGN.md 5.8) and is only in a library built with -DSOAP_EXPERIMENTS "
                    "(SOAP_BUILD_EXPERIMENTS=1 python -c 'import __graft_entry__ as g; g.build()')")
    halo_tasks.stage_chunk(engine, data_np)
    resident = 4.0 * halos_np["read_radius"] if read_factor < 2.0 else None   # (restarts and rebuilds inside the engine)
    ha1, res1, raw1, t1, _ = _run(engine, props, sc, halos_np, pipeline=False, resident=resident)
    ha2, res2, raw2, t2, _ = _run(engine, props, sc, halos_np, pipeline=4, resident=resident)
    n_team, n_failed = engine.kernel_ms("n_team"), engine.kernel_ms("n_team_failed")
    assert n_team > 0.2 * len(halos_np["index"]) and n_failed <= 0.3 * n_team, (n_team, n_failed)   # (a sphere that outgrows its arena slice is left to the single-kernel path)
    assert np.array_equal(raw1["status"], raw2["status"]) and np.array_equal(ha1["done"], ha2["done"])
    assert np.array_equal(ha1["read_radius"], ha2["read_radius"]) and np.array_equal(ha1["search_radius"], ha2["search_radius"])
    assert np.array_equal(raw1["stats"][:, [0, 1, 2, 4, 5]], raw2["stats"][:, [0, 1, 2, 4, 5]])
    bad = np.argwhere(t1 != t2)
    assert bad.size == 0, (len(bad), [(int(r), engine_columns(engine, props, int(c))) for r, c in bad[:6]])
    # ... with the tile sums and tile scans of the property stage (another order of the same additions)
    monkeypatch.setenv("SOAP_ENGINE_BPARTIALS", "1")
    ha3, _, raw3, t3, _ = _run(engine, props, sc, halos_np, pipeline=4, resident=resident)
    assert np.array_equal(raw1["status"], raw3["status"]) and np.array_equal(raw1["stats"][:, [0, 1, 2, 4, 5]], raw3["stats"][:, [0, 1, 2, 4, 5]])
    scale = np.maximum(np.abs(t1).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(t3 - t1) <= 1e-11 * np.abs(t1) + 1e-13 * scale)
    engine.close()
