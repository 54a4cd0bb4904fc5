"""Host-side mirror of the reference's operator surface: names, density bookkeeping, calc
descriptors, filters, unit factors (no GPU needed)."""

import numpy as np
import pytest

from oracle import halo_oracle as ho
from soap_amd import compute_halo_properties as chp
from soap_amd import layout
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.particle_selection import SO_properties, aperture_properties
from soap_amd.particle_selection.halo_properties import SearchRadiusTooSmallError
from soap_amd.property_table import full_property_list, stored_value
from soap_amd.synthetic import make_chunk, to_numpy
from soap_amd.units import SnapshotScalars
from tests import helpers


def test_unit_factors_are_bitwise_those_of_the_oracle():
    sc = SnapshotScalars(boxsize=215.4)
    u = helpers.oracle_units(sc)
    for r in (30.0, 50.0, 100.0, 0.5, 3000.0):
        assert sc.kpc_to_comoving(r) == u.kpc_to_comoving(r)
        assert sc.mpc_to_comoving(r) == u.mpc_to_comoving(r)
    assert sc.G_eff == u.G_eff and sc.softening_comoving() == u.softening_comoving(1)
    assert sc.virBN98 == pytest.approx(ho.flamingo_like_units().virBN98, rel=1e-15)
    for mult in (50.0, 200.0, 2500.0):
        assert sc.density_to_comoving(mult * sc.critical_density) == u.density_to_comoving(mult * u.critical_density)


def test_halo_prop_list_order_names_and_densities():
    sc = SnapshotScalars()
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((100.0, 30.0, 50.0)))
    names = [p.name for p in props]
    assert names[0] == "bound_subhalo"
    assert names[1:10] == ["SO_200_mean", "SO_50_crit", "SO_100_crit", "SO_200_crit", "SO_500_crit",
                           "SO_1000_crit", "SO_2500_crit", "SO_BN98", "SO_5xR_500_crit"]  # fmt: skip
    # apertures sorted by radius, inclusive and exclusive interleaved in dict order
    assert [p.group_name for p in props[10:]] == [
        "InclusiveSphere/30kpc", "ExclusiveSphere/30kpc", "InclusiveSphere/50kpc",
        "ExclusiveSphere/50kpc", "InclusiveSphere/100kpc", "ExclusiveSphere/100kpc"]  # fmt: skip
    so200c = props[4]
    assert so200c.group_name == "SO/200_crit" and so200c.virial_definition
    assert so200c.mean_density_multiple == 1000.0 and so200c.critical_density_multiple == 200.0
    assert props[8].group_name == "SO/BN98" and props[9].group_name == "SO/5xR_500_crit"
    # same target density as the oracle's scheduler (halo_tasks.py:306-317)
    u, oprops = helpers.oracle_prop_list(sc, cf, None, chp.aperture_variations((100.0, 30.0, 50.0)))
    assert halo_tasks.target_density_of(props, sc) == ho.target_density_of(oprops, u)
    assert [p.name for p in oprops] == names
    assert [p.group_name for p in oprops] == [p.group_name for p in props]


def test_calc_descriptors():
    sc = SnapshotScalars()
    cf = CategoryFilter(dmo=True)
    so_vars = {"500_crit": {"value": 500.0, "type": "crit", "filter": "general"},
               "5xR500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 5.0}}  # fmt: skip
    props = chp.build_halo_prop_list(sc, cf, so_vars, chp.aperture_variations((30.0, 50.0)))
    descs = [p.calc_desc(props, i) for i, p in enumerate(props)]
    assert [d.kind for d in descs] == [0, 1, 3, 4, 5, 4, 5]
    assert descs[1].filter_limit == 100 and descs[1].filter_types == 0b0010  # DMO: only DM counted
    assert descs[2].parent == 1 and descs[2].radius == 5.0
    assert descs[4].parent == -1 and descs[6].parent == 4  # exclusive 50 kpc copies from exclusive 30 kpc
    assert descs[5].parent == -1  # inclusive spheres never skip by default (compute_halo_properties.py:365-370)
    assert descs[6].prev_radius == sc.kpc_to_comoving(30.0)
    assert descs[3].physical_radius_mpc == sc.mpc_to_comoving(0.03)
    with pytest.raises(RuntimeError):
        SO_properties.RadiusMultipleSOProperties(sc, None, None, cf, "basic", 200.0, 5.0, "crit").calc_desc(props, 3)
    with pytest.raises(AttributeError):
        SO_properties.SOProperties(sc, None, None, cf, "basic", 200.0, "median")
    assert issubclass(SearchRadiusTooSmallError, Exception)


def test_category_filter_matches_reference_semantics():
    cf = CategoryFilter(dmo=False)
    hr = {"BoundSubhalo/NumberOfDarkMatterParticles": (np.array([60], dtype=np.uint32),),
          "BoundSubhalo/NumberOfGasParticles": (np.array([50], dtype=np.uint32),),
          "BoundSubhalo/NumberOfStarParticles": (np.array([3], dtype=np.uint32),),
          "BoundSubhalo/NumberOfBlackHoleParticles": (np.array([0], dtype=np.uint32),)}  # fmt: skip
    do = cf.get_do_calculation(hr)
    assert do == {"basic": True, "general": True, "dm": False, "gas": False, "star": False, "baryon": False}
    do = CategoryFilter(dmo=True).get_do_calculation(hr)
    assert do["general"] is False and do["dm"] is False
    assert CategoryFilter(dmo=False).engine_filter("baryon") == (100, 0b0101)
    assert CategoryFilter().engine_filter("basic") == (0, 0)


def test_stored_value_casts_once():
    a = 0.5
    v = stored_value("Ldm", np.array([1.0, 2.0, 1e-3]), a)
    assert v.dtype == np.float32 and np.array_equal(v, np.float32(np.array([0.5, 1.0, 5e-4])))
    assert stored_value("Ndm", np.array([12345.0]), a).dtype == np.uint32
    assert stored_value("com", np.array([1.0, 2.0, 3.0]), a).dtype == np.float64
    for key, prop in full_property_list.items():
        assert prop.shape in (1, 3, 6)


def test_synthetic_chunk_is_self_consistent():
    sc, data, halos = make_chunk(30000, 25, seed=2)
    d, h = to_numpy(data, halos)
    assert np.all(np.diff(h["nr_bound_part"]) <= 0)  # processing order: largest first
    grnr = d[1]["GroupNr_bound"]
    for row in range(len(h["index"])):
        assert (grnr == h["index"][row]).sum() == h["nr_bound_part"][row]
    # one particle sits exactly on every halo centre
    pos = d[1]["Coordinates"]
    for c in h["cofp"][:5]:
        assert np.any(np.all(pos == c, axis=1))
    assert pos.min() >= 0 and pos.max() < sc.boxsize


def test_result_set_columns_from_a_table():
    """ResultSet cuts named, typed columns out of an engine-layout table without a GPU: dataset names per
    group (and per projection), one cast per property, property filters as row masks driven by the bound
    subhalo's counts."""
    from soap_amd import compute_halo_properties as chp
    from soap_amd import layout
    from soap_amd.core.category_filter import CategoryFilter
    from soap_amd.core.result_set import ResultSet
    from soap_amd.units import SnapshotScalars

    sc = SnapshotScalars()
    cf = CategoryFilter(dmo=False)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}},
                                     chp.aperture_variations((50.0,)), projected_vars=chp.projected_variations((30.0,)))  # fmt: skip
    rs = ResultSet(props, layout.desc_columns)
    n = 7
    rng = np.random.default_rng(3)
    table = rng.uniform(1.0, 2.0, size=(n, int(rs.coloff[-1])))
    sub = layout.SUBHALO_COLS
    table[:, sub["Ngas"][0]] = [0, 150, 0, 99, 100, 5, 300]
    table[:, sub["Ndm"][0]] = 1000
    table[:, sub["Nstar"][0]] = 0
    table[:, sub["Nbh"][0]] = 0
    ha = {"index": np.arange(n) + 10, "cofp": rng.uniform(0, 1, (n, 3)), "is_central": np.ones(n, dtype=np.int32),
          "nr_bound_part": np.full(n, 1000)}  # fmt: skip
    rs.append_table(table, ha)
    cols = rs.columns()
    assert cols["SO/200_crit/SORadius"].dtype == np.float32 and cols["SO/200_crit/SORadius"].shape == (n,)
    assert cols["BoundSubhalo/CentreOfMass"].dtype == np.float64 and cols["BoundSubhalo/CentreOfMass"].shape == (n, 3)
    assert cols["BoundSubhalo/NumberOfGasParticles"].dtype == np.uint32
    assert "ProjectedAperture/30kpc/projy/TotalMass" in cols
    assert np.array_equal(cols["InputHalos/HaloCatalogueIndex"], ha["index"])
    # every property of every calculation, per projection for the projected apertures
    want = sum(len(hp.enabled_properties()) * (3 if hp.kind == layout.CALC_PROJECTED else 1) for hp in props) + 4
    assert len(cols) == want
    # the first column of the table is BoundSubhalo/TotalMass, stored as float32
    assert np.array_equal(cols["BoundSubhalo/TotalMass"], table[:, 0].astype(np.float32))


def test_property_table_rows_are_the_references():
    """Every row of soap_amd/property_table.py against the reference's table (fixture written by
    tests/golden/make_property_rows.py): output name, shape, dtype, unit, DMO flag, physical / comoving output and
    the a-scale exponent are what a SOAP catalogue carries."""
    import json
    import os

    from soap_amd.property_table import full_property_list

    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "property_rows.json")))
    assert sorted(rows) == sorted(full_property_list)
    for key, prop in full_property_list.items():
        r = rows[key]
        got = (prop.name, prop.shape, np.dtype(prop.dtype).name, prop.unit.replace(" ", ""), prop.dmo_property,
               prop.output_physical, prop.a_scale_exponent)
        want = (r["name"], r["shape"], r["dtype"], r["unit"], r["dmo_property"], r["output_physical"], r["a_scale_exponent"])
        assert got == want, (key, got, want)


def test_packed_catalogue_equals_the_columns_cut_on_the_host():
    """PackedLayout (what leaves the GPU in a multi-GPU run: stored dtypes, comoving -> stored-unit factor applied) against
    ResultSet.append_table (the host path) on a random table with the real layouts: every dataset bit for bit, same
    metadata, NumPy and torch packers identical.  soap_engine_desc_columns is a pure function: no GPU needed."""
    import torch

    from soap_amd import compute_halo_properties as chp
    from soap_amd import layout
    from soap_amd.core.category_filter import CategoryFilter
    from soap_amd.core.result_set import PackedLayout, ResultSet

    for hydro in (False, True):
        sc = SnapshotScalars()
        cf = CategoryFilter(dmo=not hydro)
        props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 100.0)),
                                         projected_vars=chp.projected_variations((30.0,)) if hydro else {})
        pk = PackedLayout(props, layout.desc_columns)
        rng = np.random.default_rng(1)
        n = 37
        table = rng.normal(size=(n, pk.table_columns)) * 10.0 ** rng.integers(-3, 6, size=(n, pk.table_columns))
        for name, kind, c0, nc, dtype, *_ in pk.fields:
            if kind == "wide" and np.issubdtype(np.dtype(dtype), np.integer):
                table[:, pk.wide_cols[c0 : c0 + nc]] = rng.integers(0, 5000, size=(n, nc))
        ha = dict(cofp=rng.uniform(0, 100, (n, 3)), index=np.arange(n), is_central=np.ones(n, dtype=np.int64), nr_bound_part=np.arange(n))
        a = ResultSet(props, layout.desc_columns)
        a.append_table(table, ha)
        narrow, wide = pk.pack(table)
        b = ResultSet(props, layout.desc_columns)
        b.append_packed(pk, narrow, wide, ha)
        ca, cb = a.columns(), b.columns()
        assert set(ca) == set(cb) and a.meta == b.meta
        for k in ca:
            assert ca[k].dtype == cb[k].dtype and ca[k].shape == cb[k].shape and np.array_equal(ca[k], cb[k]), k
        n2, w2 = pk.pack(torch.from_numpy(table))
        assert np.array_equal(n2.numpy(), narrow) and np.array_equal(w2.numpy(), wide)
        assert pk.bytes_per_halo() < 0.6 * 8 * pk.table_columns


def test_core_excised_so_variation_builds_the_reference_class():
    """An SO variation with `core_excision_fraction` becomes a CoreExcisedSOProperties entry
    (compute_halo_properties.py:271-285); its extra properties are sub-grid ones outside this engine's set, so the
    entry describes itself to the engine exactly as the plain SOProperties of the same value and type does."""
    sc = SnapshotScalars()
    cf = CategoryFilter(dmo=True)
    plain = {"500_crit": {"value": 500.0, "type": "crit"}}
    excised = {"500_crit": {"value": 500.0, "type": "crit", "core_excision_fraction": 0.15}}
    pa = chp.build_halo_prop_list(sc, cf, plain, chp.aperture_variations((50.0,)))
    pb = chp.build_halo_prop_list(sc, cf, excised, chp.aperture_variations((50.0,)))
    assert type(pb[1]) is SO_properties.CoreExcisedSOProperties and type(pa[1]) is SO_properties.SOProperties
    assert isinstance(pb[1], SO_properties.SOProperties) and pb[1].core_excision_fraction == 0.15
    assert [p.name for p in pa] == [p.name for p in pb] and [p.group_name for p in pa] == [p.group_name for p in pb]
    assert pb[1].enabled_properties() == pa[1].enabled_properties()
    da, db = pa[1].calc_desc(pa, 1), pb[1].calc_desc(pb, 1)
    assert bytes(da) == bytes(db)
    assert halo_tasks.target_density_of(pa, sc) == halo_tasks.target_density_of(pb, sc)


def test_halo_result_wraps_into_the_tuples_result_set_append_takes():
    """reference_halo_result: every entry of the facade's halo_result as the 4-tuple ResultSet.append consumes
    (SOAP/core/result_set.py:74-130): a unit-carrying array in the table's dtype and unit (x a**exponent when the stored
    value is comoving), then description, physical flag, exponent.  The unit module here is the unit-aware stand-in the
    golden vectors were made with (tests/golden/_shims_units); the checks are the ones append makes."""
    import importlib.util
    import os
    import sys

    from soap_amd.particle_selection.halo_properties import reference_halo_result

    here = os.path.join(os.path.dirname(__file__), "golden", "_shims_units", "unyt", "__init__.py")
    spec = importlib.util.spec_from_file_location("_units_standin", here)
    unyt = importlib.util.module_from_spec(spec)
    sys.modules["_units_standin"] = unyt
    spec.loader.exec_module(unyt)
    a = 0.769
    unyt.define_golden_system(a, 1.0e-3, 43.0)

    sc = SnapshotScalars(a=a)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, {"200_crit": {"value": 200.0, "type": "crit"}}, chp.aperture_variations((50.0,)))
    rng = np.random.default_rng(11)
    ncols = [int(layout.desc_columns(hp.calc_desc(props, i))) for i, hp in enumerate(props)]
    results = []
    for halo in range(3):
        halo_result = {}
        for hp, nc in zip(props, ncols):
            hp.unpack(rng.uniform(1.0, 2.0, size=nc), halo_result, None)
        results.append(reference_halo_result(halo_result, unyt, registry=None))

    # what ResultSet.append does with them (result_set.py:96-130)
    arrays = {}
    for i, res in enumerate(results):
        for name, (data, description, physical, a_exponent) in res.items():
            assert hasattr(data, "units") and isinstance(description, str)
            if name not in arrays:
                arr = np.empty_like(unyt.unyt_array(data, registry=data.units.registry), shape=(len(results),) + data.shape)
                arrays[name] = [arr, description, physical, a_exponent]
            arr = arrays[name][0]
            assert data.units == arr.units and data.dtype == arr.dtype and data.shape == arr.shape[1:], name
            arr[i, ...] = data
    r = arrays["SO/200_crit/SORadius"]
    assert r[0].dtype == np.float32 and r[0].shape == (3,) and r[2] is False and r[3] == 1
    assert r[0].units == unyt.Unit("snap_length") * unyt.Unit("a") ** 1
    # comoving -> physical is a factor a, as in the reference's registry
    assert np.allclose(np.asarray(r[0].to("snap_length")), np.asarray(r[0]) * a, rtol=1e-6)
    com = arrays["BoundSubhalo/CentreOfMass"]
    assert com[0].dtype == np.float64 and com[0].shape == (3, 3)
    n = arrays["BoundSubhalo/NumberOfDarkMatterParticles"]
    assert n[0].dtype == np.uint32 and n[0].units == unyt.Unit("dimensionless") and n[2] is True
    m = arrays["SO/200_crit/TotalMass"]
    assert m[0].units == unyt.Unit("snap_mass") and m[2] is True


def test_facades_carry_what_chunk_task_and_the_parameter_file_read():
    """`particle_properties` and `property_list` of every entry of halo_prop_list, driven the way the reference's callers
    drive them: the property union of ChunkTask.__call__ (SOAP/core/chunk_tasks.py:220-241) and the parameter-file
    validation (SOAP/core/parameter_file.py:176-195).  The union is exactly the set of datasets stage_chunk stages."""
    from soap_amd.core.dataset_names import mass_dataset, ptypes_for_so_masses

    sc = SnapshotScalars()
    for dmo in (True, False):
        cf = CategoryFilter(dmo=dmo)
        props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 50.0)),
                                         projected_vars=chp.projected_variations((30.0,)))
        # chunk_tasks.py:220-241, statement for statement
        properties = {}
        need_so = False
        for halo_prop in props:
            if halo_prop.mean_density_multiple is not None or halo_prop.critical_density_multiple is not None:
                need_so = True
        assert need_so
        for ptype in ptypes_for_so_masses:
            properties[ptype] = set(["Coordinates", mass_dataset(ptype)])
        for halo_prop in props:
            for ptype in halo_prop.particle_properties:
                if ptype not in properties:
                    properties[ptype] = set()
                properties[ptype] = properties[ptype].union(halo_prop.particle_properties[ptype])
        staged = {"Coordinates", "Velocities", "Masses", "GroupNr_bound", "FOFGroupIDs"}
        assert properties["PartType0"] == properties["PartType1"] == properties["PartType4"] == staged
        assert properties["PartType5"] == (staged - {"Masses"}) | {"DynamicalMasses"}
        assert properties["PartType6"] == {"Coordinates", "Masses", "Weights"}
        # every instance owns its lists (the reference appends to them per instance)
        props[1].particle_properties["PartType0"].append("Temperatures")
        assert "Temperatures" not in props[2].particle_properties["PartType0"]
        # parameter_file.py:176-195: names a parameter file may list for a halo type
        for hp in props:
            valid = [prop.name for prop in hp.property_list.values()]
            assert len(valid) == len(hp.property_names) and all(isinstance(n, str) and n for n in valid)
            assert set(hp.property_filters) <= set(valid)
            for key in hp.enabled_properties():
                assert hp.property_list[key].name in valid
        assert "SORadius" in [p.name for p in props[1].property_list.values()]
        assert "EncloseRadius" in [p.name for p in props[0].property_list.values()]
        by_type = {hp.base_halo_type for hp in props}
        assert by_type == {"SubhaloProperties", "SOProperties", "ApertureProperties", "ProjectedApertureProperties"}
