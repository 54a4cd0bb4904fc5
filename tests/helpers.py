"""Shared test plumbing: oracle <-> engine adapters.  The oracle (oracle/) is the checker;
it is only ever imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""

from __future__ import annotations

import numpy as np

from oracle import halo_oracle as ho
from soap_amd import compute_halo_properties as chp
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.property_table import full_property_list
from soap_amd.units import SnapshotScalars


def oracle_units(sc: SnapshotScalars) -> ho.Units:
    soft = sc.softening
    return ho.Units(
        a=sc.a, boxsize=sc.boxsize, critical_density=sc.critical_density,
        mean_density=sc.mean_density, kpc=sc.kpc, G=sc.G,
        softening={0: soft, 1: soft, 4: soft, 5: soft},
        nu_density=sc.nu_density, virBN98=sc.virBN98, H=sc.H, Omega_g=sc.Omega_g, Omega_m=sc.Omega_m,
    )  # fmt: skip


def oracle_filters(category_filter: CategoryFilter):
    short = {"BoundSubhalo/NumberOfDarkMatterParticles": "Ndm",
             "BoundSubhalo/NumberOfGasParticles": "Ngas",
             "BoundSubhalo/NumberOfStarParticles": "Nstar",
             "BoundSubhalo/NumberOfBlackHoleParticles": "Nbh"}  # fmt: skip
    return {k: dict(limit=v["limit"], properties=tuple(short[p] for p in v["properties"]))
            for k, v in category_filter.filters.items()}  # fmt: skip


def oracle_prop_list(sc, category_filter, SO_variations=None, aperture_vars=None, projected_vars=None):
    u = oracle_units(sc)
    f = oracle_filters(category_filter)
    dmo = category_filter.dmo
    classes = dict(
        subhalo=lambda: ho.SubhaloProperties(u, f, dmo),
        so=lambda filt, val, typ: ho.SOProperties(u, val, typ, halo_filter=filt, filters=f, dmo=dmo),
        so_multiple=lambda filt, val, mult, typ: ho.RadiusMultipleSOProperties(
            u, val, mult, typ, halo_filter=filt, filters=f, dmo=dmo),
        inclusive=lambda filt, r, radii: ho.ApertureProperties(
            u, r, True, radii, halo_filter=filt, filters=f, dmo=dmo),
        exclusive=lambda filt, r, radii: ho.ApertureProperties(
            u, r, False, radii, halo_filter=filt, filters=f, dmo=dmo),
        projected=lambda filt, r, radii: ho.ProjectedApertureProperties(
            u, r, radii, halo_filter=filt, filters=f, dmo=dmo),
        inclusive_property=lambda filt, prop: ho.ApertureProperties(
            u, None, True, [], halo_filter=filt, filters=f, dmo=dmo, aperture_property=prop),
        exclusive_property=lambda filt, prop: ho.ApertureProperties(
            u, None, False, [], halo_filter=filt, filters=f, dmo=dmo, aperture_property=prop),
    )  # fmt: skip
    return u, chp.build_halo_prop_list(None, None, SO_variations, aperture_vars, classes=classes,
                                       projected_vars=projected_vars)


def run_oracle(sc, data_np, halos_np, category_filter, SO_variations=None, aperture_vars=None,
               resolution=None, rows=None, projected_vars=None):  # fmt: skip
    """Oracle end to end on a NumPy chunk.  Returns (results dict row -> halo_result, stats,
    halo_arrays after the pass)."""
    u, props = oracle_prop_list(sc, category_filter, SO_variations, aperture_vars, projected_vars)
    data = {pt: d for pt, d in data_np.items()}
    mesh = {}
    for pt, d in data.items():
        n = d["Coordinates"].shape[0]
        res = ho.mesh_resolution(n) if resolution is None else resolution
        mesh[pt] = ho.Mesh(d["Coordinates"], res)
    ha = {k: v.copy() for k, v in halos_np.items()}
    ha["done"] = np.zeros(len(ha["index"]), dtype=np.int32)
    if rows is not None:
        ha["done"][:] = 1
        ha["done"][rows] = 0
    stats = []
    res = ho.process_halos(mesh, data, props, u, ha, stats)
    return dict(res), stats, ha


def engine_value(halo_result, group, key):
    """fp64 pre-cast value the engine produced for the reference's internal property key."""
    name = full_property_list[key].name
    return halo_result[f"{group}/{name}"][4]


# quantities that are sums of signed terms: compare with an absolute floor tied to the scale
# of the terms ("fp64 parity at 1e-10 despite reduction order")
ATOL_FACTOR = 1e-12   # the one absolute floor of the engine-vs-oracle comparisons: 1e-12 x (scale of the summed terms)


def assert_close(name, got, want, rtol=1e-10, scale=0.0, atol_factor=ATOL_FACTOR):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    tol = rtol * np.abs(want) + atol_factor * scale
    bad = np.abs(got - want) > tol
    assert not np.any(bad), f"{name}: engine {got} vs oracle {want} (tol {tol})"
