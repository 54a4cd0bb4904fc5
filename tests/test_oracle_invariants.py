"""
The invariants and known-answer checks the REFERENCE's own tests assert for this path,
applied to the oracle:
  tests/test_shared_mesh.py:76-124   ball query == brute-force minimum-image set
  tests/test_SO_properties.py:379-446  NFW concentration within 10 % of the input
  error contracts: test_SO_properties.py:167-184, test_aperture_properties.py:119-128
"""

import numpy as np
import pytest
from scipy.special import lambertw

from oracle import halo_oracle as ho


def _brute(pos, centre, radius, boxsize):
    dr = pos - centre[None, :]
    dr[dr > 0.5 * boxsize] -= boxsize
    dr[dr < -0.5 * boxsize] += boxsize
    return np.sum(dr**2, axis=1)


@pytest.mark.parametrize("resolution", [1, 2, 4, 8, 16, 32])
def test_mesh_query_equals_brute_force(resolution):
    rng = np.random.RandomState(resolution)
    boxsize = 1.0
    cases = [(1000, np.full(3, 0.5), 0.5, False, 0.25)]
    for box_wrap in (True, False):
        for _ in range(4):
            cases.append((1000, rng.random_sample(3) * boxsize, 0.25 * rng.random_sample() * boxsize,
                          box_wrap, None))  # fmt: skip
    cases += [(0, np.full(3, 0.5), 0.5, False, 0.25), (1, np.full(3, 0.5), 0.5, False, 0.25)]
    for npts, centre, radius, box_wrap, max_search in cases:
        pos = 2 * radius * rng.random_sample((npts, 3)) - radius + centre[None, :]
        if box_wrap:
            pos = pos % boxsize
        mesh = ho.Mesh(pos, resolution)
        for _ in range(20):
            c = rng.random_sample(3) * 2 * radius - radius + centre
            R = rng.random_sample() * (max_search if max_search else radius)
            idx = mesh.query_radius_periodic(c, R, pos, boxsize)
            assert len(idx) == len(np.unique(idx))
            if npts == 0:
                assert len(idx) == 0
                continue
            r2 = _brute(pos.copy(), c, R, boxsize)
            in_idx = np.zeros(npts, dtype=bool)
            in_idx[idx] = True
            assert not np.any(r2[in_idx] > R * R)
            assert not np.any((~in_idx) & (r2 < R * R))


def _rnfw(rng, n, con):
    p = rng.random_sample(int(n))
    p *= np.log(1.0 + con) - con / (1.0 + con)
    return (-(1.0 / np.real(lambertw(-np.exp(-p - 1)))) - 1) / con


@pytest.mark.parametrize("concentration", [5.0, 10.0])
def test_nfw_concentration_known_answer(concentration):
    """tests/test_SO_properties.py:379-446: 1e4-particle NFW halos, SO/200_crit/Concentration
    within 10 % of the input concentration (median over seeds)."""
    u = ho.flamingo_like_units()
    got = []
    for seed in range(6):
        rng = np.random.RandomState(seed)
        npart = 10000
        m_200 = 1.0e4  # 1e14 Msun
        rho200 = 200.0 * u.critical_density
        r_200 = (m_200 / (rho200 * (4 / 3) * np.pi)) ** (1 / 3) / u.a  # comoving
        radius = np.sort(r_200 * _rnfw(rng, npart, concentration))
        if radius[-1] < r_200:
            radius[-1] = r_200 * 1.01
        radius[0] = 0.0
        phi = 2.0 * np.pi * rng.random_sample(npart)
        st = 2.0 * rng.random_sample(npart) - 1.0
        ct = np.sqrt((1.0 - st) * (1.0 + st))
        centre = 100.0 * rng.random_sample(3)
        pos = centre + np.stack([radius * np.cos(phi) * st, radius * np.sin(phi) * st, radius * ct], axis=1)
        data = {1: {"Coordinates": pos, "Velocities": 1000.0 * (rng.random_sample((npart, 3)) - 0.5),
                    "Masses": np.ones(npart) * m_200 / npart,
                    "GroupNr_bound": np.ones(npart, dtype=np.int32),
                    "FOFGroupIDs": np.ones(npart, dtype=np.int64)}}  # fmt: skip
        so = ho.SOProperties(u, 200.0, "crit")
        out = so._calculate({"cofp": centre, "index": 1, "is_central": 1}, 2 * radius.max(), data)
        got.append(out["concentration_soft"])
    assert abs(np.median(got) / concentration - 1.0) < 0.1


def test_search_radius_too_small_contracts():
    u = ho.flamingo_like_units()
    rng = np.random.RandomState(3)
    n = 500
    r = np.sort(rng.exponential(1.0 / 60.0, n))
    r[0] = 0.0
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    centre = np.array([50.0, 50.0, 50.0])
    data = {1: {"Coordinates": centre + d * r[:, None], "Velocities": rng.normal(size=(n, 3)),
                "Masses": np.full(n, 5.0), "GroupNr_bound": np.ones(n, dtype=np.int32),
                "FOFGroupIDs": np.ones(n, dtype=np.int64)}}  # fmt: skip
    halo = {"cofp": centre, "index": 1, "is_central": 1, "nr_bound_part": n}
    res = {}
    ho.SubhaloProperties(u).calculate(halo, 1.0, data, res)
    # aperture larger than the search radius (test_aperture_properties.py:119-128)
    with pytest.raises(ho.SearchRadiusTooSmallError):
        ho.ApertureProperties(u, 50.0, True).calculate(halo, u.kpc_to_comoving(10.0), data, res)
    # radius multiple without its parent (test_SO_properties.py:155-165)
    with pytest.raises(RuntimeError):
        ho.RadiusMultipleSOProperties(u, 500.0, 5.0, "crit").calculate(halo, 1.0, data, res)
    # a density so low that the sphere never reaches it -> needs a larger radius
    so = ho.SOProperties(u, 200.0, "mean")
    so.reference_density = 1e-6
    with pytest.raises(ho.SearchRadiusTooSmallError):
        so.calculate(halo, 1.0, data, res)
    # BoundSubhalo gate: fewer bound particles than the catalogue says (subhalo_properties.py:2632-2646)
    with pytest.raises(ho.SearchRadiusTooSmallError):
        ho.SubhaloProperties(u).calculate(dict(halo, nr_bound_part=n + 1), 1.0, data, {})
    with pytest.raises(RuntimeError):
        ho.SubhaloProperties(u).calculate(dict(halo, nr_bound_part=n - 1), 1.0, data, {})
