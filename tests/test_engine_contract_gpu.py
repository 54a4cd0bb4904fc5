"""
GPU tests of the engine contract beyond the value parity of test_parity_gpu.py:
  * the member set the property kernels SUM OVER (the sorted slab prefix of a calculation, fetched through the
    member probe) equals the reference's selection, index for index -- north star: "SO membership indices bit-exact";
  * the STORED values (one cast to the dtype of the property table) equal the oracle's cast bit for bit or within one
    ulp of float32;
  * read-radius updates applied inside the engine (resident radius) give what the reference's repeated
    process_halos calls give;
  * device memory stays bounded when chunks of different sizes are staged one after the other.
"""

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd import layout
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.property_table import full_property_list
from soap_amd.synthetic import make_chunk, to_numpy
from tests import helpers
from tests.test_parity_gpu import _abs_scale, _scales

pytestmark = pytest.mark.gpu


def _wrapped_radius(pos, c, L):
    off = c - 0.5 * L
    rel = ((pos - off) % L) + off - c   # halo_tasks.py:113-117, then pos - centre
    return np.sqrt(np.sum(rel**2, axis=1))


def test_member_probe_returns_the_selection_of_the_reference(built_library):
    sc, data, halos = make_chunk(150000, 60, seed=15, hydro=True)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=False)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "500_crit": {"value": 500.0, "type": "crit"},
               "3xR500": {"value": 500.0, "type": "crit", "radius_multiple": 3.0}}  # fmt: skip
    ap_vars = chp.aperture_variations((50.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars)
    names = [hp.group_name for hp in props]
    ref, _, _ = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    L = sc.boxsize
    slot_of = {0: 0, 1: 1, 4: 2, 5: 3}
    nchecked = 0
    for row in (0, 1, 7, 23, 41, 59):
        c = halos_np["cofp"][row]
        hidx = int(halos_np["index"][row])
        for ci, grp in enumerate(names):
            ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
            engine.set_member_probe(row, ci)
            out = halo_tasks.process_halos(engine, props, sc, ha, [])
            assert out[2] == 0
            ts, ix = engine.fetch_members()
            want_ts, want_ix = [], []
            for pt, d in data_np.items():
                r = _wrapped_radius(d["Coordinates"], c, L)
                bound = d["GroupNr_bound"] == hidx
                if grp == "BoundSubhalo":
                    m = bound
                elif grp.startswith("SO/"):
                    R = float(ref[row][f"{grp}/r"].raw[0])
                    m = (r < R) if R > 0 else np.zeros(len(r), bool)   # strict (SO_properties.py:485)
                else:
                    R = sc.kpc_to_comoving(float(grp.rsplit("/", 1)[1][:-3]))
                    m = r <= R   # inclusive (aperture_properties.py:310)
                    if grp.startswith("ExclusiveSphere"):
                        m &= bound
                idx = np.flatnonzero(m)
                want_ts.append(np.full(len(idx), slot_of[pt], dtype=np.int32))
                want_ix.append(idx)
            want_ts, want_ix = np.concatenate(want_ts), np.concatenate(want_ix)
            if grp.startswith("ExclusiveSphere") and len(ts) == 0 and len(want_ix) > 0:
                # the copy rule (aperture_properties.py:4086-4123): the values come from the next-smaller sphere,
                # whose members are the same bound particles -- the engine records nothing for a copy
                r_prev = props[ci].all_radii_kpc[props[ci].all_radii_kpc.index(props[ci].aperture_physical_radius_kpc) - 1]
                enc32 = np.float32(ref[row]["BoundSubhalo/EncloseRadius"].stored[0])
                assert sc.kpc_to_comoving(r_prev) > float(enc32), (row, grp)
                continue
            order = np.lexsort((want_ix, want_ts))
            assert np.array_equal(ts, want_ts[order]), (row, grp)
            assert np.array_equal(ix, want_ix[order]), (row, grp)
            nchecked += len(ix)
    engine.set_member_probe(None, 0)
    engine.close()
    assert nchecked > 10000


def _ulp32(x):
    x = np.abs(np.asarray(x, dtype=np.float32))
    return np.spacing(np.maximum(x, np.finfo(np.float32).tiny)).astype(np.float64)


@pytest.mark.parametrize("hydro", [False, True])
def test_stored_values_equal_the_oracle_cast(built_library, hydro):
    """What ends up in the catalogue: float32 / uint32 / float64 after ONE cast (SO_properties.py:3682-3699).
    Counts and float64 columns within 1e-10; float32 columns bit-equal or one ulp apart -- a sum that differs in
    its last fp64 bits can land on the other side of a float32 rounding boundary."""
    sc, data, halos = make_chunk(120000, 70, seed=31, hydro=hydro)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=not hydro)
    ap_vars = chp.aperture_variations((30.0, 100.0))
    pj_vars = chp.projected_variations((30.0,)) if hydro else {}
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars, projected_vars=pj_vars)
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    results = []
    halo_tasks.process_halos(engine, props, sc, ha, results)
    engine.close()
    ref, _, _ = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars, projected_vars=pj_vars)
    # comoving -> stored unit: the rows extracted from the reference's own table (tests/golden/property_rows.json)
    # say which outputs are physical and with which power of a their comoving expression scales
    # (SO_properties.py:3693-3699 converts to the table unit, then casts): independent of soap_amd.property_table
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "property_rows.json")) as f:
        ref_rows = json.load(f)
    n32 = nexact = 0
    for row, hr in enumerate(results):
        for okey, val in ref[row].items():
            group, key = okey.rsplit("/", 1)
            prop = full_property_list[key]
            if not hydro and not prop.dmo_property:
                continue
            got = hr[f"{group}/{prop.name}"][0]
            rr = ref_rows[key]
            # power of a between the comoving expression and the stored unit: outputs stored comoving keep it (0);
            # physical outputs built from comoving lengths convert every length: angular momenta m x cross v (+1),
            # inertia tensors x x (+2; the reduced ones are dimensionless), flow rates m |v_r| / dR (-1, SO_properties.py:2906-2909)
            k = 0
            if rr["output_physical"]:
                k = 1 if key in ("Ldm", "Lgas", "Lstar", "Lbaryons") else (2 if ("InertiaTensor" in key and "snap_length" in rr["unit"]) else (-1 if key.endswith("FlowRate") else 0))
            assert k == prop.a_power, (key, k, prop.a_power)
            want = np.zeros(val.raw.shape, dtype=np.dtype(rr["dtype"]))
            if np.issubdtype(want.dtype, np.integer):
                want[:] = val.raw.astype(want.dtype)
            else:
                want += val.raw * sc.a**k if k else val.raw   # zero-initialised table-dtype array += value
            if k == 0:
                assert np.array_equal(want, val.stored, equal_nan=True), okey   # the oracle's own cast
            assert got.dtype == want.dtype == prop.dtype, okey
            if np.issubdtype(prop.dtype, np.integer):
                assert np.array_equal(got, want), okey
            elif prop.dtype == np.float64:
                assert np.allclose(got, want, rtol=1e-10, atol=1e-10 * sc.boxsize), okey
            else:
                # differences of large terms (angular momenta, dispersion off-diagonals, centre-of-mass offsets) are
                # compared on the scale of the terms, everything else to one unit in the last place of float32
                raw_scale = float(np.max(np.abs(val.raw))) if val.raw.size else 0.0
                cancelling = key.startswith(("L", "veldisp", "vcom", "com", "proj_veldisp", "kappa", "DtoT")) or "Inertia" in key
                M, vscale = _scales(ref[row], group)
                term_scale = _abs_scale(key, M, vscale) * sc.a**k   # scale of the terms of a cancelling sum (fp64 noise 1e-12 of it)
                tol = _ulp32(want) + (1e-9 * raw_scale + 1e-12 * term_scale if cancelling else 0.0)
                d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                assert np.all(d <= tol), (okey, got, want)
                n32 += got.size
                nexact += int(np.sum(got == want))
    assert n32 > 5000 and nexact > 0.98 * n32, (n32, nexact)


def test_resident_radius_replaces_the_retry_passes(built_library):
    """Halos whose calculations outgrow the read radius: with the region known to be resident the engine applies the
    reference's updates (read <- max(1.5 read, search), search <- radius reached; halo_tasks.py:167-187, :393-402)
    itself and finishes the halo in one call -- same values (to summation order), same final radii, same n_loop and
    search sphere of the final attempt as the repeated calls of ChunkTask's loop."""
    sc, data, halos = make_chunk(80000, 50, seed=7, read_radius_factor=1.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((50.0,)))
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    H = len(halos_np["index"])
    # (a) the reference's way: call again until nothing is left
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
    table_a = np.zeros((H, 0))
    stats_a = np.zeros((H, layout.STATS_COLUMNS), dtype=np.int64)
    ncall = 0
    while np.any(ha["done"] == 0) and ncall < 12:
        out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        raw = out[4]
        ok = raw["status"] == 0
        t = engine.fetch(table=True)["table"]
        if table_a.shape[1] == 0:
            table_a = np.zeros((H, t.shape[1]))
        table_a[raw["rows"][ok]] = t[ok]
        stats_a[raw["rows"][ok]] = raw["stats"][ok]
        ncall += 1
    assert ncall > 2 and np.all(ha["done"] == 1)
    # (b) one call with the whole box declared resident
    hb = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(engine, props, sc, hb, None, unpack=False, resident_radius=np.inf)
    assert out[2] == 0 and np.all(hb["done"] == 1)
    table_b = engine.fetch(table=True)["table"]
    # the same numbers up to the order of summation: a halo whose search radius grew between two calls may be handed
    # to a larger thread group the second time (the class follows a work estimate), which rounds sums differently
    scale = np.maximum(np.abs(table_a).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(table_a - table_b) <= 1e-11 * np.abs(table_a) + 1e-13 * scale)
    assert np.mean(table_a == table_b) > 0.9   # (the class limits follow the load of a call: a later call over few halos uses other groups)
    assert np.array_equal(ha["read_radius"], hb["read_radius"])
    assert np.array_equal(ha["search_radius"], hb["search_radius"])
    assert np.array_equal(stats_a[:, :2], out[4]["stats"][:, :2])
    assert np.any(hb["read_radius"] != halos_np["read_radius"])
    # (c) a resident radius that covers ONE update per call: fewer calls than (a), the same final radii
    hc = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
    ncall_c = 0
    while np.any(hc["done"] == 0) and ncall_c < 12:
        out = halo_tasks.process_halos(engine, props, sc, hc, None, unpack=False, resident_radius=1.6 * hc["read_radius"])
        ncall_c += 1
        if ncall_c == 1:
            assert out[2] > 0   # one update is not enough for this chunk
    assert 1 < ncall_c < ncall and np.all(hc["done"] == 1)
    assert np.array_equal(ha["read_radius"], hc["read_radius"])
    assert np.array_equal(ha["search_radius"], hc["search_radius"])
    engine.close()


def test_device_memory_stays_bounded_over_chunks_of_different_sizes(built_library):
    import torch

    sc, data, halos = make_chunk(300000, 100, seed=4)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, {})
    engine = halo_tasks.make_engine(sc, device=0)
    halos_np = {k: v.numpy() for k, v in halos.items()}
    d = {k: v.cuda() for k, v in data[1].items()}
    free = []
    sizes = [300000, 211111, 287654, 199999, 299990, 123457, 290001, 250003]
    for n in sizes:
        chunk = {1: {k: v[:n].contiguous() for k, v in d.items()}}
        halo_tasks.stage_chunk(engine, chunk)
        ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
        # halos of the particles that were cut off fail the bound-particle gate of the reference: only staging is at stake
        try:
            halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        except RuntimeError:
            pass
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    engine.close()
    # after the first few chunks the footprint no longer grows with every new size (it did: one cached generation
    # of ~60-110 B per particle per distinct size)
    assert free[-1] >= free[2] - 64 * 2**20, free


def test_runs_the_pipeline_cannot_serve_stay_on_the_single_kernel_path(built_library, monkeypatch):
    """The one-lane-per-halo radius loop of the pipeline (stage_a.hip) knows nothing of the inertia tensors' search-radius rule
    nor of property-defined aperture radii: a run that carries them must not enter the pipeline even when it is forced on for
    every halo (SOAP_ENGINE_PIPELINE=2) -- and still gives the answers of the default configuration."""
    sc, data, halos = make_chunk(150000, 80, seed=41)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    ap = chp.aperture_variations((50.0,))
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, ap)   # default parameters: the tensors are on
    assert any(p.calc_desc(props, i).flags & layout.CALC_FLAG_INERTIA for i, p in enumerate(props))

    def run():
        engine = halo_tasks.make_engine(sc, device=0)
        halo_tasks.stage_chunk(engine, data_np)
        ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
        out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        table = engine.fetch(table=True)["table"]
        n_pipe = int(engine.kernel_ms("n_pipeline"))
        engine.close()
        return table, out[4]["status"], n_pipe

    t0, s0, n0 = run()
    monkeypatch.setenv("SOAP_ENGINE_PIPELINE", "2")
    t2, s2, n2 = run()
    assert n0 == 0 and n2 == 0
    assert np.array_equal(s0, s2) and np.array_equal(t0, t2)


def test_stored_catalogue_packed_by_the_engine_equals_the_host_rule(built_library):
    """soap_engine_pack_catalogue: the stored columns of the result table, scaled and cast once, in the caller's row order,
    written by ONE pass over the engine's table -- bit for bit what PackedLayout.pack does with a copy of the table
    (SOAP/core/result_set.py:96-130 and SO_properties.py:3693-3699 restated) followed by a row gather."""
    import torch

    from soap_amd.core.result_set import PackedLayout

    sc, data, halos = make_chunk(120000, 80, seed=31, hydro=True)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=False)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)), projected_vars=chp.projected_variations((50.0,)))
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    H = len(halos_np["index"])
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
    halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
    table = torch.as_tensor(engine.fetch(table=True)["table"], device="cuda:0")
    packed = PackedLayout(props, engine.desc_columns)
    narrow, wide = packed.pack(table)
    order = torch.randperm(H, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    n2, w2 = packed.pack_engine(engine, "cuda:0", order=order)
    assert n2.dtype == torch.float32 and w2.dtype == torch.float64 and n2.shape == (H, narrow.shape[1]) and w2.shape == (H, wide.shape[1])
    assert torch.equal(n2, narrow[order]) and torch.equal(w2, wide[order])
    n3, w3 = packed.pack_engine(engine, "cuda:0", order=None)
    assert torch.equal(n3, narrow) and torch.equal(w3, wide)
    sub = order[: H // 3].contiguous()
    n4, w4 = packed.pack_engine(engine, "cuda:0", order=sub)
    assert torch.equal(n4, narrow[sub]) and torch.equal(w4, wide[sub])
    engine.close()


def test_the_scheduling_guess_changes_no_result(built_library, monkeypatch):
    """k_predict guesses, from a thinned copy of the particles, where a halo's radius loop will end; the first sorted superset,
    the thread-group class and the launch order follow the guess.  Whatever it says -- here it is made to speak up for every
    halo with at least one thinned particle -- selections, radius rounds, radii and statuses are those of a run without it
    and the values agree to rounding (another class sums in another order); and it says the same in every run."""
    sc, data, halos = make_chunk(300000, 200, seed=61, read_radius_factor=8.0, halo_fraction=0.7)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)))
    H = len(halos_np["index"])
    engine = halo_tasks.make_engine(sc, device=0)

    def run(predict, pred_min=None):
        monkeypatch.setenv("SOAP_ENGINE_PREDICT", "1" if predict else "0")
        monkeypatch.setenv("SOAP_ENGINE_PRED_FORCE", "1")   # (the halos of this box are isolated: none would get a guess by itself)
        monkeypatch.setenv("SOAP_ENGINE_PRED_MARGIN", "1.5")
        if pred_min is None:
            monkeypatch.delenv("SOAP_ENGINE_PRED_MIN", raising=False)
        else:
            monkeypatch.setenv("SOAP_ENGINE_PRED_MIN", str(pred_min))
        halo_tasks.stage_chunk(engine, data_np)
        ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=np.inf)
        return ha, out[4], engine.fetch(table=True)["table"], int(engine.kernel_ms("n_predicted"))

    ha0, raw0, t0, np0 = run(False)
    ha1, raw1, t1, np1 = run(True, pred_min=1)
    ha2, raw2, t2, np2 = run(True, pred_min=1)
    assert np0 == 0 and np1 > 0 and np2 == np1
    for ha, raw in ((ha1, raw1), (ha2, raw2)):
        assert np.array_equal(raw["status"], raw0["status"]) and np.array_equal(ha["done"], ha0["done"])
        assert np.array_equal(raw["stats"][:, [0, 1, 4, 5]], raw0["stats"][:, [0, 1, 4, 5]])   # rounds, sphere content, rounds of all attempts, attempts
        assert np.array_equal(ha["search_radius"], ha0["search_radius"]) and np.array_equal(ha["read_radius"], ha0["read_radius"])
    scale = np.maximum(np.abs(t0).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(t1 - t0) <= 1e-11 * np.abs(t0) + 1e-13 * scale)
    assert np.array_equal(t1, t2)   # the same guess, the same classes, the same sums: bit for bit
    engine.close()
