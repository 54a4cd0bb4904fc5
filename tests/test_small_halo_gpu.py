"""
The small-halo class (soap_amd/csrc/small_halo.hip: one wavefront per halo, the halo's slab resident in LDS) against the
oracle and against the general path (the thread-group classes of halo_kernel.inc).  Same selections -- radius rounds, sphere
contents, candidates, statuses and radii handed back are equal exactly -- and the same sums in another order.
"""

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy
from tests import helpers
from tests.test_parity_gpu import compare_halo

pytestmark = pytest.mark.gpu


class _NoInertia:
    """parameter-file stand-in: everything on except the inertia tensors (a run that asks for them stays on the general path)"""

    def get_property_filters(self, base_halo_type, names):
        return {n: (False if "InertiaTensor" in n else "basic") for n in names}

    def strict_halo_copy(self):
        return False


def _run(engine, props, sc, halos_np, small, monkeypatch, resident=None, groups=None):
    """small: 0 the general path only; 1 only the halos that are expected to fit go through the small class; 2 / 3 / 5 every halo of the
    wavefront class tries its 128-entry / 512-entry / 1024-entry variant first; 4 (the default) the variant the bound-particle count suggests (what
    does not fit is handed on to the wavefront class inside the pass)"""
    monkeypatch.setenv("SOAP_ENGINE_SMALL", str(small))
    if groups is not None:
        monkeypatch.setenv("SOAP_ENGINE_GS", str(groups))
    else:
        monkeypatch.delenv("SOAP_ENGINE_GS", raising=False)
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    results = []
    out = halo_tasks.process_halos(engine, props, sc, ha, results, resident_radius=resident)
    table = engine.fetch(table=True)["table"]
    return ha, results, out[4], table, (engine.kernel_ms("n_small"), engine.kernel_ms("n_small_escalated"))


def _same_selections(ha_a, raw_a, ha_b, raw_b):
    assert np.array_equal(raw_a["status"], raw_b["status"])
    assert np.array_equal(ha_a["done"], ha_b["done"])
    assert np.array_equal(ha_a["read_radius"], ha_b["read_radius"]) and np.array_equal(ha_a["search_radius"], ha_b["search_radius"])
    assert np.array_equal(raw_a["stats"][:, [0, 1, 2, 4, 5]], raw_b["stats"][:, [0, 1, 2, 4, 5]])   # loops, n_sel, n_cand, loops of all attempts, attempts


def _close(t_a, t_b):
    scale = np.maximum(np.maximum(np.abs(t_a), np.abs(t_b)).max(axis=0, keepdims=True), 1e-300)
    return np.all(np.abs(t_a - t_b) <= 1e-11 * np.abs(t_b) + 1e-13 * scale)


@pytest.mark.parametrize("seed,npart,nhalo,read_factor,sat", [(71, 250000, 400, 8.0, 0.0), (72, 300000, 300, 1.5, 0.15)])
def test_small_class_matches_the_oracle_and_the_general_path(built_library, monkeypatch, seed, npart, nhalo, read_factor, sat):
    sc, data, halos = make_chunk(npart, nhalo, seed=seed, satellite_fraction=sat, read_radius_factor=read_factor)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars, parameters=_NoInertia())   # bound subhalo, 9 SO definitions incl. 5 x R_500, six spheres
    H = len(halos_np["index"])
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha0, res0, raw0, t0, (n0, _) = _run(engine, props, sc, halos_np, 0, monkeypatch)
    ha1, res1, raw1, t1, (n1, e1) = _run(engine, props, sc, halos_np, 1, monkeypatch)
    assert n0 == 0 and n1 > 0.2 * H, (n0, n1)
    _same_selections(ha1, raw1, ha0, raw0)
    assert _close(t1, t0)
    # every halo of the wavefront class through the small class first: the ones that outgrow its slab are escalated
    ha2, res2, raw2, t2, (n2, e2) = _run(engine, props, sc, halos_np, 2, monkeypatch)
    assert n2 > n1 and e2 > 0, (n1, n2, e2)
    _same_selections(ha2, raw2, ha0, raw0)
    assert _close(t2, t0)
    # ... and through the variant with 512 slab entries (its streamed columns in global memory)
    ha6, res6, raw6, t6, (n6, e6) = _run(engine, props, sc, halos_np, 3, monkeypatch)
    assert n6 >= n2 and e6 < e2, (n2, n6, e2, e6)
    _same_selections(ha6, raw6, ha0, raw0)
    assert _close(t6, t0)
    # ... and through the variant with 1024 slab entries (measured, no gain: only in a library built with -DSOAP_EXPERIMENTS)
    if engine.kernel_ms("experiments") == 1.0:
        ha8, res8, raw8, t8, (n8, e8) = _run(engine, props, sc, halos_np, 5, monkeypatch)
        assert n8 >= n6 and e8 <= e6, (n6, n8, e6, e8)
        _same_selections(ha8, raw8, ha0, raw0)
        assert _close(t8, t0)
    # the default: every halo of the wavefront class through the variant its bound-particle count suggests
    ha7, res7, raw7, t7, (n7, e7) = _run(engine, props, sc, halos_np, 4, monkeypatch)
    assert n7 >= n2, (n2, n7)
    _same_selections(ha7, raw7, ha0, raw0)
    assert _close(t7, t0)
    # few persistent waves: every wave works through many halos
    ha3, _, raw3, t3, (n3, _) = _run(engine, props, sc, halos_np, 1, monkeypatch, groups=3)
    assert n3 == n1 and np.array_equal(t3, t1)
    _same_selections(ha3, raw3, ha0, raw0)
    if read_factor < 2.0:
        assert np.any(raw1["status"] == 1)   # this chunk has halos that need a larger read radius
        # with the box declared resident the retries happen inside the engine (restarts inside the small class)
        ha4, _, raw4, t4, _ = _run(engine, props, sc, halos_np, 0, monkeypatch, resident=np.inf)
        ha5, _, raw5, t5, (n5, _) = _run(engine, props, sc, halos_np, 1, monkeypatch, resident=np.inf)
        assert n5 > 0 and np.all(ha4["done"] == 1) and np.all(ha5["done"] == 1) and raw5["stats"][:, 5].max() > 1
        _same_selections(ha5, raw5, ha4, raw4)
        assert _close(t5, t4)
    engine.close()
    # and the oracle
    from oracle import halo_oracle as ho

    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    try:
        ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars)
    finally:
        ho.ENABLE_INERTIA = saved
    assert np.array_equal(ha1["done"], ha_ref["done"])
    done_rows = np.flatnonzero(ha1["done"] == 1)
    assert len(res1) == len(done_rows) == len(ref)
    for j, row in enumerate(done_rows):
        o = {k: v for k, v in ref[row].items() if "InertiaTensor" not in k}
        compare_halo(o, res1[j], sc)
        compare_halo(o, res2[j], sc)
        compare_halo(o, res6[j], sc)
    st = raw1["stats"][raw1["status"] == 0]
    assert np.array_equal(st[:, 0], [s["n_loop"] for s in stats])
    assert np.array_equal(st[:, 1], [s["n_sel"] for s in stats])
    assert np.array_equal(st[:, 2], [s["n_cand"] for s in stats])


def test_small_class_so_membership_is_exact(built_library, monkeypatch):
    """particle-in-aperture counts and SO member counts of halos done by the small class, against the oracle's index sets"""
    sc, data, halos = make_chunk(120000, 150, seed=73)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "500_crit": {"value": 500.0, "type": "crit"}, "200_mean": {"value": 200.0, "type": "mean"}}
    ap_vars = chp.aperture_variations((30.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha, res, raw, table, (ns, _) = _run(engine, props, sc, halos_np, 2, monkeypatch)
    assert ns > 0.5 * len(halos_np["index"])
    from oracle import halo_oracle as ho

    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    try:
        ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
    finally:
        ho.ENABLE_INERTIA = saved
    done_rows = np.flatnonzero(ha["done"] == 1)
    assert len(done_rows) == len(ref) > 0
    for j, row in enumerate(done_rows):
        for okey, val in ref[row].items():
            group, key = okey.rsplit("/", 1)
            if key == "Ndm":
                assert np.array_equal(helpers.engine_value(res[j], group, key), val.raw), (row, okey)
    engine.close()


def test_small_class_inertia_tensors_equal_the_general_path(built_library, monkeypatch):
    """the tensor pass of the small-halo kernels (lanes = tensors for shapes and eigenproblems, lanes = particles for the sweeps)
    against the thread-group classes: same selections, same iterations -- the tensors agree to rounding -- for the default list
    (bound subhalo + 9 SOs, 40 tensors per halo) and with halos that are handed on"""
    sc, data, halos = make_chunk(250000, 300, seed=74, satellite_fraction=0.1)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)))   # default parameters: the tensors are on
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha0, res0, raw0, t0, (n0, _) = _run(engine, props, sc, halos_np, 0, monkeypatch)
    ha4, res4, raw4, t4, (n4, e4) = _run(engine, props, sc, halos_np, 4, monkeypatch)
    ha2, res2, raw2, t2, (n2, e2) = _run(engine, props, sc, halos_np, 2, monkeypatch)   # 128-entry slabs for everybody: many are handed on
    engine.close()
    assert n0 == 0 and n4 > 0.5 * len(halos_np["index"]) and e2 > 0
    for ha, raw, t in ((ha4, raw4, t4), (ha2, raw2, t2)):
        _same_selections(ha, raw, ha0, raw0)
        scale = np.maximum(np.maximum(np.abs(t), np.abs(t0)).max(axis=0, keepdims=True), 1e-300)
        assert np.all(np.abs(t - t0) <= 1e-9 * np.abs(t0) + 1e-12 * scale)
    ntens = 0
    for r in res4:
        for grp in ("BoundSubhalo", "SO/200_crit", "SO/2500_crit", "SO/5xR_500_crit"):
            it = r[f"{grp}/TotalInertiaTensor"][4]
            ntens += int(np.any(it))
            assert np.array_equal(r[f"{grp}/DarkMatterInertiaTensor"][4], it)
    assert ntens > 300


def test_small_class_inertia_tensors_with_restarts_inside_the_resident_radius(built_library, monkeypatch):
    """halos whose read radius grows inside the engine (resident region) start again with every calculation open: the tensor
    states of the small-halo kernels are rebuilt with them"""
    sc, data, halos = make_chunk(300000, 300, seed=72, read_radius_factor=1.5, satellite_fraction=0.15)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0,)))   # the default SO list (5 x R_500 reaches far out), tensors on
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha0, _, raw0, t0, (n0, _) = _run(engine, props, sc, halos_np, 0, monkeypatch, resident=np.inf)
    ha4, _, raw4, t4, (n4, _) = _run(engine, props, sc, halos_np, 4, monkeypatch, resident=np.inf)
    engine.close()
    assert n0 == 0 and n4 > 0 and np.all(ha4["done"] == 1) and raw4["stats"][:, 5].max() > 1   # some halos needed more than one attempt
    _same_selections(ha4, raw4, ha0, raw0)
    scale = np.maximum(np.maximum(np.abs(t4), np.abs(t0)).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(t4 - t0) <= 1e-9 * np.abs(t0) + 1e-12 * scale)


def test_escalation_without_the_device_hand_over_list(built_library, monkeypatch):
    """SOAP_ENGINE_SMALL_HANDOVER=0: halos that outgrow the small-halo class' slab are not handed on inside the pass; they
    come back with the overflow status and the host escalates them to the wavefront class in a pass of its own (the
    `next[0]` path of soap_engine_run).  The switch is read on every run, so one process can compare both ways: same
    selections, same table."""
    sc, data, halos = make_chunk(250000, 300, seed=73, read_radius_factor=8.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)), parameters=_NoInertia())
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha_a, _, raw_a, t_a, (n_a, e_a) = _run(engine, props, sc, halos_np, 2, monkeypatch)
    passes_a = engine.kernel_ms("n_passes")
    monkeypatch.setenv("SOAP_ENGINE_SMALL_HANDOVER", "0")
    ha_b, _, raw_b, t_b, (n_b, e_b) = _run(engine, props, sc, halos_np, 2, monkeypatch)
    passes_b = engine.kernel_ms("n_passes")
    monkeypatch.delenv("SOAP_ENGINE_SMALL_HANDOVER")
    assert e_a > 0 and e_b == e_a and n_b == n_a, (n_a, e_a, n_b, e_b)
    assert passes_a == 1 and passes_b == 2, (passes_a, passes_b)   # the escalated halos needed a second launch of the wavefront class
    _same_selections(ha_a, raw_a, ha_b, raw_b)
    assert np.array_equal(t_a, t_b)
    assert np.all(raw_b["status"] >= 0)
    engine.close()
