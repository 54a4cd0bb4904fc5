"""
Parity of the HIP engine against the CPU oracle on identical synthetic chunks
(-m gpu: runs on the MI355X box, through the C ABI).

Bars (BASELINE.json north_star): particle counts bit-exact, SO membership indices
bit-exact, fp64 values within 1e-10 relative.  Sums of signed terms (angular momentum,
dispersion off-diagonals, centre-of-mass offsets) additionally get an absolute floor of
1e-12 x (scale of the summed terms) -- `helpers.ATOL_FACTOR`, the one value every comparison
here uses -- because NumPy's pairwise summation and the device's tree reduction round
differently.
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

pytestmark = pytest.mark.gpu

COUNT_KEYS = ("Ndm", "Ngas", "Nstar", "Nbh", "Nnu")


def _scales(o, group):
    """Scale of the summed terms for the cancelling quantities of one selection."""
    def g(k):
        v = o.get(f"{group}/{k}")
        return None if v is None else v.raw
    M = max(float(np.abs(g("Mtot")).max()) if g("Mtot") is not None else 0.0, 1e-300)
    vd = g("veldisp_matrix_dm")
    sig2 = float(np.abs(vd[:3]).max()) if vd is not None else 0.0
    vc = g("vcom")
    v2 = float(np.abs(vc).max()) ** 2 if vc is not None else 0.0
    vscale = np.sqrt(max(sig2, v2, 1.0))
    return M, vscale


def compare_halo(o, hr, sc, rtol=1e-10, dmo=True):
    nchecked = 0
    for okey, val in o.items():
        group, key = okey.rsplit("/", 1)
        if dmo and not full_property_list[key].dmo_property:
            # a DMO run does not emit it (SO_properties.py:3607-3608); the oracle keeps a zero
            assert f"{group}/{full_property_list[key].name}" not in hr
            continue
        got = helpers.engine_value(hr, group, key)
        want = val.raw
        nchecked += 1
        if key in COUNT_KEYS:
            assert np.array_equal(got, want), f"{okey}: {got} != {want}"
            continue
        M, vscale = _scales(o, group)
        R = sc.boxsize  # generous length scale for com offsets
        scale = _abs_scale(key, M, vscale)
        if "InertiaTensor" in key:   # off-diagonal terms are differences: compare on the scale of the tensor
            scale = float(np.abs(want).max())
        helpers.assert_close(okey, got, want, rtol=rtol, scale=scale)
    return nchecked


def _abs_scale(key, M, vscale):
    """Absolute tolerance scale of quantities that are differences of large terms (components of L or of
    a dispersion tensor can be tiny next to the terms they are the difference of)."""
    if key in ("Ldm", "Lgas", "Lstar", "Lbaryons"):
        return M * 2.0 * vscale
    if key.startswith("veldisp_matrix_"):
        return vscale**2
    if key.startswith("vcom") or key.startswith("proj_veldisp_"):
        return vscale
    if key.startswith("com") or key.startswith("DtoT"):
        return 1.0
    return 0.0


def _run_engine(sc, data_np, halos_np, props, **engine_kw):
    engine = halo_tasks.make_engine(sc, device=0, **engine_kw)
    halo_tasks.stage_chunk(engine, data_np)
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    results = []
    out = halo_tasks.process_halos(engine, props, sc, ha, results)
    return engine, ha, results, out


@pytest.mark.parametrize("seed,npart,nhalo", [(3, 60000, 40), (11, 200000, 150)])
def test_dmo_full_set_matches_oracle(built_library, seed, npart, nhalo):
    sc, data, halos = make_chunk(npart, nhalo, seed=seed)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars)
    assert np.array_equal(ha["done"], ha_ref["done"])
    raw = out[4]
    done_rows = np.flatnonzero(ha["done"] == 1)
    assert len(results) == len(done_rows) == len(ref)
    for j, row in enumerate(done_rows):
        compare_halo(ref[row], results[j], sc)
    # scheduler bookkeeping: same number of search-radius iterations, same sphere contents
    st = raw["stats"][raw["status"] == 0]
    assert np.array_equal(st[:, 0], [s["n_loop"] for s in stats])
    assert np.array_equal(st[:, 1], [s["n_sel"] for s in stats])
    assert np.array_equal(st[:, 2], [s["n_cand"] for s in stats])
    engine.close()


def test_so_membership_indices_bit_exact(built_library):
    sc, data, halos = make_chunk(120000, 80, seed=5)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "500_crit": {"value": 500.0, "type": "crit"}}
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, _, _ = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, {})
    pos = data_np[1]["Coordinates"]
    L = sc.boxsize
    for row in range(len(halos_np["index"])):
        for grp in ("SO/200_crit", "SO/500_crit"):
            R_o = float(ref[row][f"{grp}/r"].raw[0])
            R_e = float(helpers.engine_value(results[row], grp, "r")[0])
            assert R_e == pytest.approx(R_o, rel=1e-12)
            if R_o == 0:
                continue
            c = halos_np["cofp"][row]
            off = c - 0.5 * L
            rel = ((pos - off) % L) + off - c
            r = np.sqrt(np.sum(rel**2, axis=1))
            want = np.flatnonzero(r < R_o)
            got = engine.membership(row, 1, R_e, strict=True)
            assert np.array_equal(got, want), f"halo row {row} {grp}: membership differs"
    engine.close()


def test_read_radius_retry_matches_oracle(built_library):
    """Halos whose region is too small are not finished and get the reference's updated
    radii (halo_tasks.py:169-181, :393-402); a second pass finishes them."""
    sc, data, halos = make_chunk(60000, 40, seed=7, read_radius_factor=1.0)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_mean": {"value": 200.0, "type": "mean"}}
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    u, oprops = helpers.oracle_prop_list(sc, cf, so_vars, {})
    from oracle import halo_oracle as ho

    mesh = {1: ho.Mesh(data_np[1]["Coordinates"], ho.mesh_resolution(len(data_np[1]["Masses"])))}
    ha_o = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha_e = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    npass = 0
    while np.any(ha_o["done"] == 0) and npass < 12:
        ho.process_halos(mesh, data_np, oprops, u, ha_o)
        halo_tasks.process_halos(engine, props, sc, ha_e, [])
        assert np.array_equal(ha_e["done"], ha_o["done"])
        assert np.array_equal(ha_e["read_radius"], ha_o["read_radius"])
        assert np.array_equal(ha_e["search_radius"], ha_o["search_radius"])
        npass += 1
    assert npass > 1, "test chunk should need at least one retry"
    assert np.all(ha_e["done"] == 1)
    engine.close()


def test_satellites_and_filters(built_library):
    """Satellites get zero SO columns and no density loop (halo_tasks.py:381,
    SO_properties.py:3627); filtered calculations stay zero (category_filter.py:69-106)."""
    sc, data, halos = make_chunk(80000, 60, seed=9, satellite_fraction=0.3)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit", "filter": "general"},
               "500_crit": {"value": 500.0, "type": "crit"}}  # fmt: skip
    ap_vars = chp.aperture_variations((50.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
    assert np.array_equal(ha["done"], ha_ref["done"])
    nsat = 0
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        compare_halo(ref[row], results[j], sc)
        if halos_np["is_central"][row] == 0:
            nsat += 1
            assert helpers.engine_value(results[j], "SO/500_crit", "r")[0] == 0.0
    assert nsat > 0
    engine.close()


def test_small_slab_triggers_retry_path(built_library):
    """A slab smaller than the biggest search sphere must be grown transparently."""
    sc, data, halos = make_chunk(60000, 40, seed=3)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}}
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    e1, ha1, r1, o1 = _run_engine(sc, data_np, halos_np, props, slab_capacity=256, n_workgroups=16)
    e2, ha2, r2, o2 = _run_engine(sc, data_np, halos_np, props)
    assert np.all(ha1["done"] == 1)
    # different thread-group sizes sum in different orders: equal to rounding, counts exactly
    t1, t2 = o1[4]["table"], o2[4]["table"]
    assert np.allclose(t1, t2, rtol=1e-11, atol=1e-9)
    assert np.array_equal(t1[:, 6], t2[:, 6]) and np.array_equal(t1[:, layout.NCOLS[layout.CALC_SUBHALO] + 3], t2[:, layout.NCOLS[layout.CALC_SUBHALO] + 3])
    assert np.array_equal(o1[4]["stats"][:, :3], o2[4]["stats"][:, :3])   # column 3 is a clock
    e1.close()
    e2.close()


def test_stored_dtypes_follow_the_property_table(built_library):
    sc, data, halos = make_chunk(40000, 20, seed=13)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}},
                                     chp.aperture_variations((50.0,)))  # fmt: skip
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    by_name = {p.name: p for p in full_property_list.values()}
    for hr in results:
        for key, val in hr.items():
            if key.startswith("InputHalos/"):
                continue
            prop = by_name[key.rsplit("/", 1)[1]]
            stored = val[0]
            assert stored.dtype == prop.dtype and stored.shape == (prop.shape,)
            assert val[2] == prop.output_physical
    engine.close()


def test_projected_apertures_match_oracle(built_library):
    """ProjectedAperture/{10,30,50,100}kpc/proj{x,y,z}: bound particles inside a 2D radius
    (projected_aperture_properties.py:98-198), incl. the copy rule for apertures beyond the
    enclose radius (:1827-1884)."""
    sc, data, halos = make_chunk(150000, 120, seed=21, satellite_fraction=0.2)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}}
    pj_vars = chp.projected_variations((10.0, 30.0, 100.0, 1000.0, 3000.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, chp.aperture_variations((50.0,)), projected_vars=pj_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, chp.aperture_variations((50.0,)),
                                            projected_vars=pj_vars)  # fmt: skip
    assert np.array_equal(ha["done"], ha_ref["done"])
    ncopy = 0
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        compare_halo(ref[row], results[j], sc)
        a = helpers.engine_value(results[j], "ProjectedAperture/1000kpc/projx", "Mtot")
        b = helpers.engine_value(results[j], "ProjectedAperture/3000kpc/projx", "Mtot")
        ncopy += int(a[0] == b[0] and a[0] > 0)
    assert ncopy > 0
    engine.close()


@pytest.mark.parametrize("engine_kw", [{}, {"slab_capacity": 256, "n_workgroups": 8}, {"slab_capacity": 256, "n_workgroups": 8, "f32_valued": True}],
                         ids=["default-classes", "small-slab-forces-256-and-1024-thread-groups", "float32-valued-masses-exact-running-sums"])
def test_hydro_chunk_matches_oracle(built_library, engine_kw):
    """Four particle types (gas, DM, stars, BHs) with unequal masses: type-major concatenation of
    the reference (SO_properties.py:3520-3550) only shows up as summation order; counts and masses
    per type, the sequential cumulative profile (unequal masses!) and the SO radii must agree, and so
    must the gas / star / baryon kinematics (centres of mass, angular momenta, dispersion tensors,
    half-mass radii, projected dispersions).  The second parametrisation shrinks the slabs so that the
    halos run in the 256- and 1024-thread groups (segment reductions, multi-wave scans, pipelined slab
    build) instead of the wavefront class."""
    sc, data, halos = make_chunk(160000, 100, seed=33, hydro=True)
    data_np, halos_np = to_numpy(data, halos)
    engine_kw = dict(engine_kw)
    f32_valued = engine_kw.pop("f32_valued", False)
    if f32_valued:
        # Masses as SWIFT snapshots hold them: float32 values (here in float64 arrays).  Every partial sum of the cumulative mass is
        # then exact in fp64, and the thread-group classes sum the chunks of the slab build in parallel instead of along one chain --
        # the same numbers as numpy.cumsum to the bit, or the SO radii and memberships below would not agree
        for d in data_np.values():
            d["Masses"] = d["Masses"].astype(np.float32).astype(np.float64)
    cf = CategoryFilter(dmo=False)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "500_crit": {"value": 500.0, "type": "crit"},
               "BN98": {"value": 0.0, "type": "BN98"}}  # fmt: skip
    ap_vars = chp.aperture_variations((30.0, 100.0))
    pj_vars = chp.projected_variations((30.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props, **engine_kw)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars, projected_vars=pj_vars)
    assert np.array_equal(ha["done"], ha_ref["done"])
    if engine_kw:
        assert engine.kernel_ms("n_wg256") + engine.kernel_ms("n_wg1024") > 0
        assert (engine.phase_cycles()["exact_cumsum_chunks"] > 0) == f32_valued
    nchecked = 0
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        o, hr = ref[row], results[j]
        for okey, val in o.items():
            group, key = okey.rsplit("/", 1)
            got = helpers.engine_value(hr, group, key)
            want = val.raw
            nchecked += 1
            if key in ("Ndm", "Ngas", "Nstar", "Nbh"):
                assert np.array_equal(got, want), f"{okey}: {got} != {want}"
            else:
                M, vscale = _scales(o, group)
                scale = _abs_scale(key, M, vscale)
                if "InertiaTensor" in key:
                    scale = float(np.abs(want).max())
                helpers.assert_close(okey, got, want, rtol=1e-10, scale=scale)
    assert nchecked > 5000
    ngas = sum(int(helpers.engine_value(r, "BoundSubhalo", "Ngas")[0]) for r in results)
    assert ngas > 0
    # ordered rotation (kinematic_properties.py:266-425): kappa_corot in [0, 1], disc fractions in [-1, 1], and
    # they are really computed (not all zero / one) in every class that carries them
    nrot = 0
    for r in results:
        for grp in ("BoundSubhalo", "InclusiveSphere/100kpc", "ExclusiveSphere/30kpc"):
            for key in ("kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons"):
                k = float(helpers.engine_value(r, grp, key)[0])
                assert 0.0 <= k <= 1.0 + 1e-12
                nrot += int(0.0 < k < 1.0)
        for grp in ("BoundSubhalo", "InclusiveSphere/100kpc", "SO/200_crit"):
            for key in ("DtoTgas", "DtoTstar"):
                d = float(helpers.engine_value(r, grp, key)[0])
                assert -1.0 - 1e-12 <= d <= 1.0 + 1e-12
                nrot += int(-1.0 < d < 1.0 and d != 0.0)
    assert nrot > 500
    # gas and stellar inertia tensors exist where the component has at least 20 particles
    ntens = sum(int(np.any(helpers.engine_value(r, grp, key)))
                for r in results for grp in ("BoundSubhalo", "SO/200_crit")
                for key in ("GasInertiaTensor", "StellarInertiaTensorReduced", "GasInertiaTensorNoniterative"))
    assert ntens > 100
    # 2D tensors of the projected apertures, per projection axis
    npj = sum(int(np.any(r[f"ProjectedAperture/100kpc/{proj}/{key}"][4]))
              for r in results for proj in ("projx", "projy", "projz")
              for key in ("ProjectedTotalInertiaTensor", "ProjectedGasInertiaTensorReduced", "ProjectedStellarInertiaTensorNoniterative"))
    assert npj > 300
    engine.close()


def test_per_halo_calculate_protocol(built_library):
    """The reference's operator protocol: prop.calculate(input_halo, search_radius, data,
    halo_result) ADDS '<group>/<PropertyName>' entries, leaves its inputs alone, and raises
    SearchRadiusTooSmallError for a halo that needs a larger region -- served from the engine's batch run."""
    from soap_amd.particle_selection.halo_properties import SearchRadiusTooSmallError

    # read radii equal to the search radii and a 200-mean sphere: some halos need a larger region
    sc, data, halos = make_chunk(60000, 40, seed=7, read_radius_factor=1.0)
    data_np, halos_np = to_numpy(data, halos)
    halos_np["read_radius"] = halos_np["read_radius"].copy()
    halos_np["read_radius"][5:] *= 8.0          # ... but only the first five
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_mean": {"value": 200.0, "type": "mean"}},
                                     chp.aperture_variations((50.0,)))  # fmt: skip
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    batch = halo_tasks.bind_results(engine, props, out[4], ha)
    done_rows = np.flatnonzero(ha["done"] == 1)
    assert 0 < len(done_rows) < len(ha["done"])
    for j, row in enumerate(done_rows[:10]):
        input_halo = {k: halos_np[k][row] for k in ("index", "cofp", "search_radius", "is_central", "nr_bound_part")}
        before = {k: np.copy(v) for k, v in input_halo.items()}
        halo_result = {}
        for hp in props:
            hp.calculate(input_halo, input_halo["search_radius"], data_np, halo_result)
        for k, v in before.items():
            assert np.array_equal(input_halo[k], v)   # inputs untouched (test_SO_properties.py:226-228)
        want = {k: v for k, v in results[j].items() if not k.startswith("InputHalos/")}
        assert set(halo_result) == set(want)
        for k in want:
            assert np.array_equal(halo_result[k][0], want[k][0]) and halo_result[k][0].dtype == want[k][0].dtype
    todo_row = int(np.flatnonzero(ha["done"] == 0)[0])
    with pytest.raises(SearchRadiusTooSmallError):
        props[1].calculate({"index": halos_np["index"][todo_row]}, 0.0, data_np, {})
    engine.close()


def test_device_resident_float32_inputs(built_library):
    """Real snapshots hold Velocities/Masses in float32 and the chunk may already live in HBM: float32 POINTERS go
    through the C ABI (`value_bytes` = 4 of soap_engine_stage_particles, widened on the device by the placement
    pass), as device tensors and as host arrays, and give one table; the same values passed as host float64 arrays give the float64
    semantics of the reference (no cast of the cumulative mass).  A float32 mass array beside float64 velocities is widened on the way
    in (one width per call: the reference would concatenate such a chunk to float64 too)."""
    import torch

    from soap_amd import engine as engine_mod

    sc, data, halos = make_chunk(50000, 30, seed=17)
    data_np, halos_np = to_numpy(data, halos)
    for d in data_np.values():
        d["Velocities"] = d["Velocities"].astype(np.float32).astype(np.float64)
        d["Masses"] = d["Masses"].astype(np.float32).astype(np.float64)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}},
                                     chp.aperture_variations((50.0,)))  # fmt: skip
    e1, ha1, r1, o1 = _run_engine(sc, data_np, halos_np, props)
    data_dev, data_h32, data_mixed = {}, {}, {}
    for pt, d in data_np.items():
        data_dev[pt] = {
            "Coordinates": torch.from_numpy(d["Coordinates"]).cuda(),
            "Velocities": torch.from_numpy(d["Velocities"].astype(np.float32)).cuda(),
            "Masses": torch.from_numpy(d["Masses"].astype(np.float32)).cuda(),
            "GroupNr_bound": torch.from_numpy(d["GroupNr_bound"]).cuda(),
            "FOFGroupIDs": torch.from_numpy(d["FOFGroupIDs"]).cuda(),
        }
        data_h32[pt] = dict(d, Velocities=d["Velocities"].astype(np.float32), Masses=d["Masses"].astype(np.float32))
        data_mixed[pt] = dict(d, Masses=d["Masses"].astype(np.float32))
    # what crosses the boundary: 4-byte values when both arrays are float32, 8-byte otherwise
    widths = []
    real = engine_mod.Engine.stage_particles

    def spy(self, type_slot, pos, vel, mass, grnr, fof, resolution=None):
        widths.append(np.dtype(engine_mod._value_dtype(vel, mass)).itemsize)
        return real(self, type_slot, pos, vel, mass, grnr, fof, resolution)

    engine_mod.Engine.stage_particles = spy
    try:
        e2, ha2, r2, o2 = _run_engine(sc, data_dev, halos_np, props)
        del data_dev   # (the staging call has consumed the arrays: include/soap_engine.h, lifetime rule)
        e3, ha3, r3, o3 = _run_engine(sc, data_h32, halos_np, props)
        e4, ha4, r4, o4 = _run_engine(sc, data_mixed, halos_np, props)
    finally:
        engine_mod.Engine.stage_particles = real
    assert widths == [4, 4, 8]
    for o in (o2, o3, o4):
        assert np.array_equal(o1[4]["status"], o[4]["status"])
    # 8-byte values at the boundary: the table of the float64 arrays, bit for bit; 4-byte values, from the device or from the host: one
    # table, bit for bit -- and the SO mass profile then goes through the reference's float32 casts of the cumulative mass
    # (SO_properties.py:400-402; pinned by tests/golden/control_flow_f32.npz in tests/test_control_flow_gpu.py), so an SO radius may be the
    # neighbouring float32 of the float64 run's and what hangs on it moves by that much
    assert np.array_equal(o1[4]["table"], o4[4]["table"])
    assert np.array_equal(o2[4]["table"], o3[4]["table"])
    scale = np.maximum(np.abs(o1[4]["table"]).max(axis=0, keepdims=True), 1e-300)
    assert np.all(np.abs(o2[4]["table"] - o1[4]["table"]) <= 2e-6 * np.abs(o1[4]["table"]) + 1e-7 * scale)
    for e in (e1, e2, e3, e4):
        e.close()


def test_chunk_harness_columns_match_per_halo_results(built_library, tmp_path):
    """run_chunk + ResultSet (the counterpart of ChunkTask / ResultSet / combine_chunks ordering): the
    vectorised columns must be exactly what the per-halo dictionaries hold, halos that needed a larger read
    radius are finished by a later pass, and the written catalogue is in (cell, catalogue index) order."""
    from soap_amd.core.result_set import run_chunk

    sc, data, halos = make_chunk(80000, 60, seed=9, read_radius_factor=1.0, hydro=True)
    data_np, halos_np = to_numpy(data, halos)
    halos_np["read_radius"] = halos_np["read_radius"].copy()
    halos_np["read_radius"][10:] *= 8.0
    cf = CategoryFilter(dmo=False)
    so_vars = {"200_mean": {"value": 200.0, "type": "mean"}}
    props = chp.build_halo_prop_list(sc, cf, so_vars, chp.aperture_variations((50.0,)),
                                     projected_vars=chp.projected_variations((30.0,)))  # fmt: skip
    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data_np)
    ha = {k: v.copy() for k, v in halos_np.items()}
    rs = run_chunk(engine, props, sc, ha)
    assert np.all(ha["done"] == 1) and rs.n_halo == len(ha["index"])
    # reference path: per-halo dictionaries, same passes
    ha2 = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    results = []
    for _ in range(8):
        if not np.any(ha2["done"] == 0):
            break
        halo_tasks.process_halos(engine, props, sc, ha2, results)
    cols = rs.columns()
    by_index = {int(r["InputHalos/HaloCatalogueIndex"][0]): r for r in results}
    assert len(by_index) == rs.n_halo
    for row, ix in enumerate(cols["InputHalos/HaloCatalogueIndex"]):
        r = by_index[int(ix)]
        assert set(r) == set(cols)
        for name, val in r.items():
            got = np.atleast_1d(cols[name][row])
            assert got.dtype == np.asarray(val[0]).dtype, name
            assert np.array_equal(got, np.atleast_1d(val[0])), name
    # writer: sorted by (cell, index), names and dtypes as the reference's datasets
    cell = sc.boxsize / 4.0
    rs.write(str(tmp_path / "cat"), cell_size=[cell] * 3, dimension=[4, 4, 4])
    z = np.load(str(tmp_path / "cat.npz"))
    assert "SO|200_mean|SORadius" in z.files and z["SO|200_mean|SORadius"].dtype == np.float32
    ci = (z["InputHalos|HaloCentre"] // cell).astype(np.int64)
    key = ci[:, 0] * 16 + ci[:, 1] * 4 + ci[:, 2]
    assert np.all(np.diff(key) >= 0)
    engine.close()


def test_many_calculations_per_family(built_library):
    """More calculations in one family than the engine finalises in one batch (PREFB = 12): the nine default
    SO definitions plus five inclusive spheres = 14 nested selections over all particles, 1 + 5 over the
    bound ones; 20 calculations in total."""
    sc, data, halos = make_chunk(90000, 50, seed=29)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    ap_vars = chp.aperture_variations((10.0, 30.0, 50.0, 100.0, 300.0))
    props = chp.build_halo_prop_list(sc, cf, None, ap_vars)
    assert 20 <= len(props) <= 32
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, None, ap_vars)
    assert np.array_equal(ha["done"], ha_ref["done"]) and ha["done"].sum() > 10
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        compare_halo(ref[row], results[j], sc)
    engine.close()


@pytest.mark.parametrize("small", ["0", "4"], ids=["thread-group-classes", "small-halo-class"])
def test_production_sized_calculation_lists(built_library, monkeypatch, small):
    """The lists the reference's production parameter files build hold 30 (FLAMINGO) and 32 (COLIBRE: 1 + 4 + 18 + 9)
    calculations (compute_halo_properties.py:232-505); the engine takes up to 64.  A COLIBRE-shaped list of 32 and a list of 45
    -- thirteen SO definitions, sixteen spheres of each kind -- against the oracle, on the thread-group classes and through the
    small-halo class."""
    monkeypatch.setenv("SOAP_ENGINE_SMALL", small)
    from oracle import halo_oracle as ho
    from tests.test_small_halo_gpu import _NoInertia

    sc, data, halos = make_chunk(90000, 60, seed=31)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so13 = {f"{v}_crit": {"value": float(v), "type": "crit"} for v in (50, 100, 200, 300, 500, 750, 1000, 1500, 2500)}
    so13.update({"200_mean": {"value": 200.0, "type": "mean"}, "BN98": {"value": 0.0, "type": "BN98"},
                 "3xR_500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 3.0}, "5xR_500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 5.0}})
    cases = [(dict(list(so13.items())[:4] + [("BN98", so13["BN98"])]), (1.0, 3.0, 5.0, 10.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0, 80.0, 100.0, 150.0), 32),
             (so13, (1.0, 3.0, 5.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0, 80.0, 100.0, 150.0, 300.0, 500.0), 46)]
    saved = ho.ENABLE_INERTIA
    ho.ENABLE_INERTIA = False
    try:
        for so_vars, radii, want_n in cases:
            ap_vars = chp.aperture_variations(radii)
            props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, parameters=_NoInertia())
            assert len(props) == want_n, len(props)
            engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
            if small != "0":
                assert engine.kernel_ms("n_small") > 0
            ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
            assert np.array_equal(ha["done"], ha_ref["done"]) and ha["done"].sum() > 10
            for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
                compare_halo({k: v for k, v in ref[row].items() if "InertiaTensor" not in k}, results[j], sc)
            st = out[4]["stats"][out[4]["status"] == 0]
            assert np.array_equal(st[:, 0], [s["n_loop"] for s in stats]) and np.array_equal(st[:, 1], [s["n_sel"] for s in stats])
            engine.close()
    finally:
        ho.ENABLE_INERTIA = saved


def _tiny_periodic_chunk(L=2.0, n=6000, centres=((0.05, 0.1, 1.95), (1.0, 1.0, 1.0)), rb=0.25, seed=0):
    """Uniform particles in a box so small that search spheres are larger than half the box: the ball query
    needs several periodic copies per dimension (shared_mesh.py:146-179)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.0, L, size=(n, 3))
    vel = rng.normal(0.0, 100.0, size=(n, 3))
    mass = np.full(n, 1.0e-2)
    grnr = np.full(n, -1, dtype=np.int32)
    index = []
    for h, c in enumerate(centres):
        d = pos - np.asarray(c)
        d -= L * np.round(d / L)
        r = np.sqrt((d**2).sum(axis=1))
        sel = (r < rb) & (grnr < 0)
        grnr[sel] = 7 + h
        index.append(7 + h)
    data = {1: {"Coordinates": pos, "Velocities": vel, "Masses": mass, "GroupNr_bound": grnr,
                "FOFGroupIDs": np.where(grnr >= 0, 1, 2147483647).astype(np.int64)}}  # fmt: skip
    nb = np.array([(grnr == ix).sum() for ix in index], dtype=np.int64)
    halos = {"index": np.array(index, dtype=np.int64), "cofp": np.array(centres, dtype=np.float64),
             "search_radius": np.full(len(index), 0.3), "read_radius": np.full(len(index), 1.9),
             "is_central": np.ones(len(index), dtype=np.int32), "nr_bound_part": nb}  # fmt: skip
    return data, halos


def test_periodic_copies_in_a_tiny_box(built_library):
    import dataclasses

    from soap_amd.units import SnapshotScalars

    sc = dataclasses.replace(SnapshotScalars(), boxsize=2.0)
    data_np, halos_np = _tiny_periodic_chunk(L=sc.boxsize)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_mean": {"value": 200.0, "type": "mean"}}
    ap_vars = chp.aperture_variations((100.0, 1000.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
    assert np.array_equal(ha["done"], ha_ref["done"])
    assert np.array_equal(ha["read_radius"], ha_ref["read_radius"]) and np.array_equal(ha["search_radius"], ha_ref["search_radius"])
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        compare_halo(ref[row], results[j], sc)
    engine.close()


def test_fatal_and_empty_cases(built_library):
    """'Found more particles than expected' aborts the chunk with the reference's message
    (subhalo_properties.py:2642-2646, halo_tasks.py:140-146); an empty halo table and an absent particle
    type are fine (SharedMesh.empty, shared_mesh.py:25-27)."""
    sc, data, halos = make_chunk(40000, 20, seed=41)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, {})
    engine = halo_tasks.make_engine(sc, device=0)
    empty = {k: v[:0] for k, v in data_np[1].items()}
    halo_tasks.stage_chunk(engine, {1: data_np[1], 0: empty})
    # no halos at all
    ha0 = dict({k: v[:0].copy() for k, v in halos_np.items()}, done=np.zeros(0, dtype=np.int32))
    res0 = []
    out0 = halo_tasks.process_halos(engine, props, sc, ha0, res0)
    assert res0 == [] and out0[2] == 0 and out0[3] == 0
    # one halo claims fewer bound particles than it has
    bad = {k: v.copy() for k, v in halos_np.items()}
    bad["nr_bound_part"][3] -= 1
    ha = dict(bad, done=np.zeros(len(bad["index"]), dtype=np.int32))
    with pytest.raises(RuntimeError, match="more particles than expected"):
        halo_tasks.process_halos(engine, props, sc, ha, [])
    engine.close()


def _dummy_generator_style_chunk(mean_density, nhalo=100, seed=4251, spacing=8.0):
    """Halos in the style of the reference's test generator (tests/dummy_halo_generator.py:751-1423, as
    summarised in SURVEY.md section 4): exponential radii with scale 1/60 Mpc around a first particle at r = 0,
    masses U[0.1, 0.5], velocities U[-500, 500], types gas/DM/star/BH with p = (0.2, 0.4, 0.39, 0.01), 60 % of the
    particles bound to the halo, ~10 % unbound, the rest bound to other objects (1 % satellites of the same FOF
    group).  One isolated halo per lattice site, so every halo sees only its own particles -- the situation
    of the reference's per-halo unit tests."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(nhalo ** (1.0 / 3.0)))
    L = spacing * side
    parts = {t: {"Coordinates": [], "Velocities": [], "Masses": [], "GroupNr_bound": [], "FOFGroupIDs": []} for t in (0, 1, 4, 5)}
    halos = {"index": [], "cofp": [], "search_radius": [], "read_radius": [], "is_central": [], "nr_bound_part": []}
    for h in range(nhalo):
        centre = (np.array([h % side, (h // side) % side, h // (side * side)], dtype=np.float64) + 0.5) * spacing
        centre += rng.uniform(-0.3, 0.3, 3)
        n = int(rng.choice([1, 10, 100, 400, 1500]))
        r = rng.exponential(1.0 / 60.0, n)
        r[0] = 0.0
        phi, ct = rng.uniform(0, 2 * np.pi, n), rng.uniform(-1, 1, n)
        st = np.sqrt(1 - ct * ct)
        pos = centre + r[:, None] * np.stack([st * np.cos(phi), st * np.sin(phi), ct], axis=1)
        pos %= L
        vel = rng.uniform(-500.0, 500.0, (n, 3))
        mass = rng.uniform(0.1, 0.5, n)
        types = rng.choice([0, 1, 4, 5], size=n, p=[0.2, 0.4, 0.39, 0.01])
        index = 1000 + 3 * h
        u = rng.uniform(0, 1, n)
        grnr = np.where(u < 0.6, index, np.where(u < 0.7, -1, np.where(u < 0.71, index + 1, index + 2))).astype(np.int32)
        grnr[0] = index
        fof = np.where(grnr == index + 2, 500000 + h, np.where(grnr >= 0, 7000 + h, 2147483647)).astype(np.int64)
        for t in (0, 1, 4, 5):
            m = types == t
            parts[t]["Coordinates"].append(pos[m]); parts[t]["Velocities"].append(vel[m]); parts[t]["Masses"].append(mass[m])
            parts[t]["GroupNr_bound"].append(grnr[m]); parts[t]["FOFGroupIDs"].append(fof[m])
        halos["index"].append(index); halos["cofp"].append(centre % L)
        halos["search_radius"].append(0.05); halos["read_radius"].append(3.9)
        halos["is_central"].append(1 if rng.uniform() < 0.9 else 0)
        halos["nr_bound_part"].append(int((grnr == index).sum()))
    # a smooth dark-matter background at the mean density: without it an isolated blob has no SO radius at all
    # (the enclosed density never drops below the threshold at a particle)
    nbg = 60000
    parts[1]["Coordinates"].append(rng.uniform(0.0, L, (nbg, 3)))
    parts[1]["Velocities"].append(rng.uniform(-500.0, 500.0, (nbg, 3)))
    parts[1]["Masses"].append(np.full(nbg, mean_density * L**3 / nbg))
    parts[1]["GroupNr_bound"].append(np.full(nbg, -1, dtype=np.int32))
    parts[1]["FOFGroupIDs"].append(np.full(nbg, 2147483647, dtype=np.int64))
    data = {t: {k: np.concatenate(v) if len(v) else np.zeros(0) for k, v in d.items()} for t, d in parts.items()}
    hal = {"index": np.array(halos["index"], dtype=np.int64), "cofp": np.array(halos["cofp"]),
           "search_radius": np.array(halos["search_radius"]), "read_radius": np.array(halos["read_radius"]),
           "is_central": np.array(halos["is_central"], dtype=np.int32),
           "nr_bound_part": np.array(halos["nr_bound_part"], dtype=np.int64)}  # fmt: skip
    order = np.argsort(-hal["nr_bound_part"], kind="stable")
    return L, data, {k: v[order] for k, v in hal.items()}


def test_reference_style_random_halos(built_library):
    """The reference's own unit tests push ~100 random multi-type halos through every SO flavour, a radius
    multiple, spheres and projected apertures (tests/test_SO_properties.py:16-376 and friends) and check
    presence / dtype / zeros for filtered halos.  Same population here, checked value by value against the
    oracle, plus those schema checks."""
    import dataclasses

    from soap_amd.units import SnapshotScalars

    sc0 = SnapshotScalars()
    L, data_np, halos_np = _dummy_generator_style_chunk(sc0.mean_density * sc0.a**3)
    sc = dataclasses.replace(sc0, boxsize=L)
    cf = CategoryFilter(dmo=False)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "200_mean": {"value": 200.0, "type": "mean"},
               "BN98": {"value": 0.0, "type": "BN98"}, "500_crit": {"value": 500.0, "type": "crit"},
               "5xR_500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 5.0}}  # fmt: skip
    ap_vars = chp.aperture_variations((30.0, 50.0))
    pj_vars = chp.projected_variations((30.0,))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars, projected_vars=pj_vars)
    assert np.array_equal(ha["done"], ha_ref["done"]) and ha["done"].sum() > 60
    nzero_filtered = 0
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        o, hr = ref[row], results[j]
        for okey, val in o.items():
            group, key = okey.rsplit("/", 1)
            prop = full_property_list[key]
            got_stored = hr[f"{group}/{prop.name}"][0]
            assert got_stored.dtype == prop.dtype and got_stored.shape == (prop.shape,)   # test_SO_properties.py:230-241
            got, want = helpers.engine_value(hr, group, key), val.raw
            if key in ("Ndm", "Ngas", "Nstar", "Nbh"):
                assert np.array_equal(got, want), f"{okey}: {got} != {want}"
            else:
                M, vscale = _scales(o, group)
                helpers.assert_close(okey, got, want, rtol=1e-10, scale=_abs_scale(key, M, vscale))
        # satellites get no SO (SO_properties.py:3600-3627): all zero
        if halos_np["is_central"][row] == 0:
            assert all(np.all(v[0] == 0) for k, v in hr.items() if k.startswith("SO/"))
            nzero_filtered += 1
    assert nzero_filtered > 0
    engine.close()


@pytest.mark.parametrize("concentration", [5.0, 10.0])
def test_nfw_concentration_known_answer_through_the_engine(built_library, concentration):
    """The reference's only physics known-answer test (tests/test_SO_properties.py:379-446): NFW halos of
    10^4 particles, SO/200_crit/Concentration within 10 % of the input -- here six halos in one chunk through the
    HIP engine, and the same numbers as the oracle to 1e-6 (the reference rounds the concentration to float32)."""
    import dataclasses

    from tests.test_oracle_invariants import _rnfw
    from soap_amd.units import SnapshotScalars

    sc = dataclasses.replace(SnapshotScalars(), boxsize=100.0)
    npart, m_200 = 10000, 1.0e4
    rho200 = 200.0 * sc.critical_density
    r_200 = (m_200 / (rho200 * (4 / 3) * np.pi)) ** (1 / 3) / sc.a
    pos, vel, grnr, cofp = [], [], [], []
    for seed in range(6):
        rng = np.random.RandomState(seed)
        radius = np.sort(r_200 * _rnfw(rng, npart, concentration))
        if radius[-1] < r_200:
            radius[-1] = r_200 * 1.01
        radius[0] = 0.0
        phi = 2.0 * np.pi * rng.random_sample(npart)
        st = 2.0 * rng.random_sample(npart) - 1.0
        ct = np.sqrt((1.0 - st) * (1.0 + st))
        centre = np.array([10.0 + 15.0 * seed, 50.0, 50.0]) + rng.random_sample(3)
        pos.append(centre + np.stack([radius * np.cos(phi) * st, radius * np.sin(phi) * st, radius * ct], axis=1))
        vel.append(1000.0 * (rng.random_sample((npart, 3)) - 0.5))
        grnr.append(np.full(npart, 10 + seed, dtype=np.int32))
        cofp.append(centre)
    n = 6 * npart
    data_np = {1: {"Coordinates": np.concatenate(pos) % sc.boxsize, "Velocities": np.concatenate(vel),
                   "Masses": np.full(n, m_200 / npart), "GroupNr_bound": np.concatenate(grnr),
                   "FOFGroupIDs": np.ones(n, dtype=np.int64)}}  # fmt: skip
    halos_np = {"index": (np.arange(6) + 10).astype(np.int64), "cofp": np.array(cofp), "search_radius": np.full(6, 0.5 * r_200),
                "read_radius": np.full(6, 6.0), "is_central": np.ones(6, dtype=np.int32),
                "nr_bound_part": np.full(6, npart, dtype=np.int64)}  # fmt: skip
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}}
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, {})
    assert np.all(ha["done"] == 1) and np.all(ha_ref["done"] == 1)
    got = np.array([helpers.engine_value(r, "SO/200_crit", "concentration_soft")[0] for r in results])
    want = np.array([ref[i]["SO/200_crit/concentration_soft"].raw[0] for i in range(6)])
    assert np.allclose(got, want, rtol=1e-6)
    assert abs(np.median(got) / concentration - 1.0) < 0.1
    engine.close()


def test_full_size_invariants_c3(built_library):
    """BASELINE.json's metric configuration (C3: 1e8 particles, 1e5 halos, 16 calculations) is too large for
    the oracle; at that size the engine is held to size-independent properties: bound-particle counts add up
    to the catalogue's, every SO obeys M = 4/3 pi R^3 rho_ref, radii and masses are ordered the way their
    thresholds are, nested spheres have nested counts, a second run reproduces the table bit for bit."""
    _full_size_invariants("c3")


def test_full_size_invariants_c2(built_library):
    """configs[1] of BASELINE.json (C2: 1e7 particles, 1e4 halos, BoundSubhalo + SO_200_crit) at full size."""
    _full_size_invariants("c2")


def test_full_size_invariants_giant_halos(built_library):
    """2e7 particles with the halo-size cap lifted to 1e6 particles (the C3 calculation list): the largest halos go
    through the 1024-thread groups with slabs of more than a million entries."""
    _full_size_invariants("giant")


def _full_size_invariants(workload):
    import torch

    import bench

    import time as _t
    _t0 = _t.time()
    def _lap(msg):
        print(f"[full-size {workload}] {msg}: {_t.time() - _t0:.1f} s", flush=True)
    wl = bench.WORKLOADS[workload]
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", max_halo=wl.get("max_halo"))
    cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
    hn = {k: v.cpu().numpy() for k, v in halos.items()}
    H = len(hn["index"])
    _lap("chunk generated")
    engine = halo_tasks.make_engine(sc, device=0)
    tables = []
    for rep in range(2):
        halo_tasks.stage_chunk(engine, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        tables.append(engine.fetch(table=True))
        _lap(f"run {rep} fetched")
    t, st = tables[0]["table"], tables[0]["status"]
    assert np.array_equal(st, tables[1]["status"]) and np.array_equal(t, tables[1]["table"])
    _lap("runs compared")
    ok = st == 0
    assert ok.sum() > 0.99 * H and np.all(st >= 0)
    coloff = np.concatenate([[0], np.cumsum([layout.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)])])
    blk = {hp.name: t[ok][:, coloff[i] : coloff[i + 1]] for i, hp in enumerate(props)}
    sub = blk["bound_subhalo"]
    S, SO, AP = layout.SUBHALO_COLS, layout.SO_COLS, layout.APERTURE_COLS
    # counts: exact
    assert np.array_equal(sub[:, S["Ndm"][0]].astype(np.int64), hn["nr_bound_part"][ok])
    assert np.all(sub[:, S["Ngas"][0]] == 0) and np.all(sub[:, S["Nstar"][0]] == 0)
    central = hn["is_central"][ok] == 1
    # every SO: M = 4/3 pi R^3 rho (SO_properties.py:209-214) wherever it exists; satellites have none
    by_name = {hp.name: hp for hp in props}
    radii = {}
    for name, hp in by_name.items():
        if hp.kind != layout.CALC_SO_DENSITY:
            continue
        b = blk[name]
        R, M = b[:, SO["r"][0]], b[:, SO["Mtot"][0]]
        assert np.all(R[~central] == 0)
        ex = R > 0
        rho = hp.calc_desc(props, props.index(hp)).reference_density
        assert np.allclose(M[ex], 4.0 / 3.0 * np.pi * R[ex] ** 3 * rho, rtol=1e-12)
        radii[rho] = R
    rhos = sorted(radii)
    for lo, hi in zip(rhos[:-1], rhos[1:]):
        both = (radii[lo] > 0) & (radii[hi] > 0)
        assert np.all(radii[lo][both] >= radii[hi][both])   # lower threshold, larger sphere
    # nested inclusive spheres: counts and masses grow with the radius; exclusive <= inclusive
    incl = sorted((hp.calc_desc(props, props.index(hp)).radius, hp.name) for hp in props if hp.kind == layout.CALC_APERTURE_INCL)
    excl = sorted((hp.calc_desc(props, props.index(hp)).radius, hp.name) for hp in props if hp.kind == layout.CALC_APERTURE_EXCL)
    for (r1, n1), (r2, n2) in zip(incl[:-1], incl[1:]):
        assert np.all(blk[n1][:, AP["Ndm"][0]] <= blk[n2][:, AP["Ndm"][0]])
        assert np.all(blk[n1][:, AP["Mtot"][0]] <= blk[n2][:, AP["Mtot"][0]] * (1 + 1e-12))
    for (ri, ni), (re_, ne) in zip(incl, excl):
        assert ri == re_
        e, i_ = blk[ne][:, AP["Ndm"][0]], blk[ni][:, AP["Ndm"][0]]
        assert np.all(e <= i_)
        assert np.all(e <= sub[:, S["Ndm"][0]])
    _lap("checks done")
    engine.close()
    del data
    torch.cuda.empty_cache()


def test_full_size_invariants_c5_hydro(built_library):
    """The hydro configuration of BASELINE.json (C5, one GPU's share: 2.5e7 particles of four types, 2.5e4 halos,
    24 calculations incl. projected apertures) at full size: the per-type bound counts add up to the catalogue's
    numbers, the per-type masses add up to the totals, a projected aperture never holds more than the bound
    subhalo, a second run reproduces the table bit for bit."""
    import torch

    import bench

    wl = bench.WORKLOADS["c5"]
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=True)
    cf, so_vars, ap_vars, pj_vars, props = bench.build_props(sc, wl)
    hn = {k: v.cpu().numpy() for k, v in halos.items()}
    H = len(hn["index"])
    engine = halo_tasks.make_engine(sc, device=0)
    tables = []
    for rep in range(2):
        halo_tasks.stage_chunk(engine, data)
        ha = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(H, dtype=np.int32))
        halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False)
        tables.append(engine.fetch(table=True))
    t, st = tables[0]["table"], tables[0]["status"]
    assert np.array_equal(st, tables[1]["status"]) and np.array_equal(t, tables[1]["table"])
    ok = st == 0
    assert ok.sum() > 0.99 * H and np.all(st >= 0)
    coloff = np.concatenate([[0], np.cumsum([layout.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)])])
    S, AP, PJ = layout.SUBHALO_COLS, layout.APERTURE_COLS, layout.PROJECTED_COLS
    sub = t[ok][:, coloff[0] : coloff[1]]
    ntot = sum(sub[:, S[k][0]] for k in ("Ngas", "Ndm", "Nstar", "Nbh"))
    assert np.array_equal(ntot.astype(np.int64), hn["nr_bound_part"][ok])
    assert sub[:, S["Ngas"][0]].sum() > 0 and sub[:, S["Nstar"][0]].sum() > 0
    msum = sum(sub[:, S[k][0]] for k in ("Mgas", "Mdm", "Mstar", "Mbh_dynamical"))
    assert np.allclose(msum, sub[:, S["Mtot"][0]], rtol=1e-12)
    # half-mass radii lie inside the bound subhalo; baryons between nothing and everything
    enc = sub[:, S["EncloseRadius"][0]]
    for k in ("HalfMassRadiusTot", "HalfMassRadiusDM", "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon"):
        assert np.all(sub[:, S[k][0]] <= enc * (1 + 1e-12)), k
    for i, hp in enumerate(props):
        b = t[ok][:, coloff[i] : coloff[i + 1]]
        if hp.kind in (layout.CALC_APERTURE_INCL, layout.CALC_APERTURE_EXCL):
            m4 = sum(b[:, AP[k][0]] for k in ("Mgas", "Mdm", "Mstar", "Mbh_dynamical"))
            assert np.allclose(m4, b[:, AP["Mtot"][0]], rtol=1e-12)
        if hp.kind == layout.CALC_PROJECTED:
            for p in range(3):
                o = p * layout.PROJECTED_NCOLS
                n4 = sum(b[:, o + PJ[k][0]] for k in ("Ngas", "Ndm", "Nstar", "Nbh"))
                assert np.all(n4 <= ntot)
    engine.close()
    del data
    torch.cuda.empty_cache()


@pytest.mark.parametrize("engine_kw", [{}, {"slab_capacity": 256, "n_workgroups": 8}], ids=["default-classes", "small-slab"])
def test_neutrino_particles_in_the_SO_profile(built_library, engine_kw):
    """PartType6: weighted masses (some negative) take part in the SO mass profile, together with a mean
    neutrino density, and feed NumberOfNeutrinoParticles / RawNeutrinoMass / NoiseSuppressedNeutrinoMass and
    the concentration; apertures, the bound subhalo and the enclosed-density test of the radius loop ignore
    them (SO_properties.py:380-395, :507-511, :2689-2753; halo_tasks.py:85-103)."""
    import dataclasses

    sc, data, halos = make_chunk(120000, 80, seed=61)
    sc = dataclasses.replace(sc, nu_density=0.05 * sc.mean_density)
    data_np, halos_np = to_numpy(data, halos)
    rng = np.random.default_rng(5)
    n_nu = 60000
    pos_nu = rng.uniform(0.0, sc.boxsize, (n_nu, 3))
    # half of the neutrino particles follow the halos
    hsel = rng.integers(0, len(halos_np["index"]), n_nu // 2)
    pos_nu[: n_nu // 2] = (halos_np["cofp"][hsel] + rng.normal(0.0, 0.3, (n_nu // 2, 3))) % sc.boxsize
    mean_m = float(np.mean(data_np[1]["Masses"]))
    data_np[6] = {"Coordinates": pos_nu, "Masses": rng.uniform(0.05, 0.15, n_nu) * mean_m,
                  "Weights": rng.normal(0.2, 0.6, n_nu)}  # fmt: skip
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "200_mean": {"value": 200.0, "type": "mean"},
               "500_crit": {"value": 500.0, "type": "crit"}, "BN98": {"value": 0.0, "type": "BN98"},
               "5xR_500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 5.0}}  # fmt: skip
    ap_vars = chp.aperture_variations((50.0, 100.0))
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars)
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props, **engine_kw)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, ap_vars)
    assert np.array_equal(ha["done"], ha_ref["done"]) and ha["done"].sum() > 40
    assert np.array_equal(ha["read_radius"], ha_ref["read_radius"]) and np.array_equal(ha["search_radius"], ha_ref["search_radius"])
    nnu = 0
    for j, row in enumerate(np.flatnonzero(ha["done"] == 1)):
        compare_halo(ref[row], results[j], sc)
        # NumberOfNeutrinoParticles is not a dark-matter-only-run property in the reference's table: the masses are
        assert "SO/200_mean/NumberOfNeutrinoParticles" not in results[j]
        nnu += int(helpers.engine_value(results[j], "SO/200_mean", "Mnu")[0] > 0)
    assert nnu > 30
    engine.close()


class _FlowRateOff:
    """Parameter-file stand-in that disables DarkMatterMassFlowRate (core/parameter_file.py:85-154)."""

    def get_property_filters(self, base_halo_type, names):
        return {n: (False if n == "DarkMatterMassFlowRate" else "basic") for n in names}


def test_dark_matter_mass_flow_rates(built_library):
    """DarkMatterMassFlowRate (SO_properties.py:2816-2932): three shells per virial SO, inflow and outflow;
    the outer shell reaches 1.05 R_SO, so the calculation may ask for a larger sphere than the SO itself."""
    sc, data, halos = make_chunk(250000, 120, seed=21)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "200_mean": {"value": 200.0, "type": "mean"},
               "BN98": {"value": 0.0, "type": "BN98"}, "500_crit": {"value": 500.0, "type": "crit"}}  # fmt: skip
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    assert [p.calc_desc(props, i).flags & layout.CALC_FLAG_DM_FLOW for i, p in enumerate(props)] == [0, 1, 1, 1, 0]
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, {})
    assert np.array_equal(ha["done"], ha_ref["done"])
    done_rows = np.flatnonzero(ha["done"] == 1)
    n_flow = 0
    for j, row in enumerate(done_rows):
        compare_halo(ref[row], results[j], sc)
        for grp in ("SO/200_crit", "SO/200_mean", "SO/BN98"):
            fl = helpers.engine_value(results[j], grp, "DarkMatterMassFlowRate")
            assert fl.shape == (6,) and np.all(fl >= 0)
            n_flow += int(np.count_nonzero(fl) == 6)
        assert np.all(helpers.engine_value(results[j], "SO/500_crit", "DarkMatterMassFlowRate") == 0)   # not a virial definition
    assert n_flow > 50
    raw = out[4]
    n_loop = raw["stats"][raw["status"] == 0][:, 0].copy()
    assert np.array_equal(n_loop, [s["n_loop"] for s in stats])
    engine.close()
    # with the property switched off in the parameter file the SOs alone decide the radius: no halo needs more
    # rounds, some need fewer, and every other column is unchanged
    props_off = chp.build_halo_prop_list(sc, cf, so_vars, {}, parameters=_FlowRateOff())
    assert all(p.calc_desc(props_off, i).flags & layout.CALC_FLAG_DM_FLOW == 0 for i, p in enumerate(props_off))
    engine, ha2, results2, out2 = _run_engine(sc, data_np, halos_np, props_off)
    raw2 = out2[4]
    n_loop2 = raw2["stats"][raw2["status"] == 0][:, 0]
    if np.array_equal(ha["done"], ha2["done"]):
        assert np.all(n_loop2 <= n_loop) and np.any(n_loop2 < n_loop)
    for j in range(min(len(results), len(results2))):
        assert "SO/200_crit/DarkMatterMassFlowRate" not in results2[j]
        assert np.array_equal(results2[j]["SO/200_crit/SORadius"][4], results[j]["SO/200_crit/SORadius"][4])
    engine.close()


class _InertiaOff:
    def get_property_filters(self, base_halo_type, names):
        return {n: (False if "InertiaTensor" in n else "basic") for n in names}


@pytest.mark.parametrize("small", ["0", "4"], ids=["thread-group-classes", "small-halo-class"])
def test_inertia_tensors(built_library, monkeypatch, small):
    """Total / dark matter inertia tensors of the bound subhalo and of SOs (inertia_tensors.py:19-132):
    iterative ellipsoids that reach into the particles surrounding the SO, fixed-sphere versions, reduced
    versions; fewer than 20 particles give zeros.  Switched off in the parameter file, their 96 columns per
    calculation disappear from the table and the other columns keep their values."""
    sc, data, halos = make_chunk(250000, 120, seed=29)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=True)
    so_vars = {"200_crit": {"value": 200.0, "type": "crit"}, "50_crit": {"value": 50.0, "type": "crit"},
               "5xR_200_crit": {"value": 200.0, "type": "crit", "radius_multiple": 5.0}}  # fmt: skip
    props = chp.build_halo_prop_list(sc, cf, so_vars, {})
    assert all(p.calc_desc(props, i).flags & layout.CALC_FLAG_INERTIA for i, p in enumerate(props))
    monkeypatch.setenv("SOAP_ENGINE_SMALL", small)   # "4": the halos of the wavefront class go through the small-halo kernels' own tensor pass
    engine, ha, results, out = _run_engine(sc, data_np, halos_np, props)
    assert (engine.kernel_ms("n_small") > 0) == (small == "4")
    ref, stats, ha_ref = helpers.run_oracle(sc, data_np, halos_np, cf, so_vars, {})
    assert np.array_equal(ha["done"], ha_ref["done"])
    done_rows = np.flatnonzero(ha["done"] == 1)
    n_full = n_moved = n_zero = 0
    for j, row in enumerate(done_rows):
        compare_halo(ref[row], results[j], sc)
        for grp in ("BoundSubhalo", "SO/200_crit", "SO/50_crit", "SO/5xR_200_crit"):
            it = helpers.engine_value(results[j], grp, "TotalInertiaTensor")
            fx = helpers.engine_value(results[j], grp, "TotalInertiaTensorNoniterative")
            rd = helpers.engine_value(results[j], grp, "TotalInertiaTensorReduced")
            dm = helpers.engine_value(results[j], grp, "DarkMatterInertiaTensor")
            assert np.array_equal(dm, it)   # dark-matter-only chunk
            if not np.any(it):
                n_zero += 1
                continue
            n_full += 1
            T = np.array([[it[0], it[3], it[4]], [it[3], it[1], it[5]], [it[4], it[5], it[2]]])
            assert np.all(np.linalg.eigvalsh(T) > 0)
            if np.any(rd):
                assert abs(rd[0] + rd[1] + rd[2] - 1.0) < 1e-9   # reduced tensor: unit trace
            n_moved += int(np.any(it != fx))
    assert n_full > 200 and n_moved > 150
    raw = out[4]
    assert np.array_equal(raw["stats"][raw["status"] == 0][:, 0], [s["n_loop"] for s in stats])
    ncols_on = engine.row_columns
    engine.close()
    props_off = chp.build_halo_prop_list(sc, cf, so_vars, {}, parameters=_InertiaOff())
    # like with like: a run with the tensors stays on the single-kernel path, so does this one (in the giants' mode the
    # large halos of so small a chunk would go through the 1024-thread pipeline kernels: same values to 1e-11, other sums)
    monkeypatch.setenv("SOAP_ENGINE_PIPELINE", "0")
    engine, ha2, results2, out2 = _run_engine(sc, data_np, halos_np, props_off)
    assert engine.row_columns == ncols_on - 96 * len(props_off)
    for j in range(len(results2)):
        assert not any("InertiaTensor" in k for k in results2[j])
        for k, v in results2[j].items():
            # a radius-multiple SO whose radius is a hair above the search radius is re-evaluated in a larger
            # sphere when the tensors are on (inertia_tensors.py:110-111): its interpolated mass may differ
            if not k.startswith("SO/5xR_200_crit/"):
                assert np.array_equal(v[0], results[j][k][0]), k
    engine.close()
