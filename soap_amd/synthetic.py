"""
Synthetic DMO / hydro chunks for parity tests and bench.py.

The reference has no offline-runnable data set (tests/run_small_volume.sh downloads
one), so every BASELINE config is a synthetic periodic box built by this recipe:

  * H halo centres uniform in the box; halo sizes from dN/dM ~ M^-1.9 scaled so
    that sum(N_h) ~ halo_fraction * N; the rest is a uniform background.
  * each halo is an NFW sphere (c in [5,10]) sampled out to 2 R200c with ONE
    particle exactly at the centre (the reference assumes a particle at the centre
    of potential: SO_properties.py:413-416; its own generator does the same,
    tests/dummy_halo_generator.py:666-667).
  * GroupNr_bound = halo index for particles within the analytic R200c, -1 else;
    FOFGroupIDs = halo index for every halo particle, NO_FOF for the background;
    nr_bound_part counted exactly (the BoundSubhalo gate requires it,
    subhalo_properties.py:2632-2646); search_radius = 1.01 * max bound radius
    (catalogue_readers/read_hbtplus.py:319-321).
  * hydro: particle types drawn with p=(0.2,0.4,0.39,0.01) for types (0,1,4,5)
    (tests/dummy_halo_generator.py:920-927).
  * particles are stored in coarse top-level-cell order (like a SWIFT snapshot,
    which is written cell by cell: SOAP/core/swift_cells.py:480-534), random
    within a cell.

Written against torch so the same code generates 1e5 particles on the CPU for tests
and 1e8 on the GPU for bench.py.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Dict

import numpy as np
import torch

NO_FOF = 2147483647
HYDRO_TYPES = (0, 1, 4, 5)
HYDRO_PROBS = (0.2, 0.4, 0.39, 0.01)


from .units import SnapshotScalars  # noqa: E402


def _nfw_table(device, n=16384, smax=20.0):
    s = torch.cat(
        [torch.zeros(1, dtype=torch.float64), torch.logspace(-5, math.log10(smax), n - 1, dtype=torch.float64)]
    ).to(device)
    m = torch.log1p(s) - s / (1.0 + s)
    return s, m


def _interp(xq, xp, fp):
    i = torch.searchsorted(xp, xq).clamp(1, xp.numel() - 1)
    x0, x1 = xp[i - 1], xp[i]
    f0, f1 = fp[i - 1], fp[i]
    t = (xq - x0) / (x1 - x0)
    return f0 + t * (f1 - f0)


def make_chunk(
    n_part: int,
    n_halo: int,
    seed: int = 0,
    hydro: bool = False,
    halo_fraction: float = 0.6,
    device: str = "cpu",
    satellite_fraction: float = 0.0,
    scalars: SnapshotScalars | None = None,
    read_radius_factor: float = 8.0,
    nmin: int = 50,
    topcells: int = 16,
    max_halo: int | None = None,
    float32_masses: bool = False,
):
    """
    Returns (scalars, data, halos):
      data[ptype] = dict(Coordinates (n,3) f64, Velocities (n,3) f64, Masses (n) f64,
                         GroupNr_bound (n) i32, FOFGroupIDs (n) i64)   [torch, on `device`]
      halos = dict(index i64, cofp (H,3) f64, search_radius f64, read_radius f64,
                   is_central i32, nr_bound_part i64)   sorted by nr_bound_part descending
                   (SOAP/core/chunk_tasks.py:118-120)
    """
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    sc = scalars or SnapshotScalars()
    if scalars is None:
        # keep the particle mass fixed (2.44e9 Msun, ~FLAMINGO-like): L ~ N^(1/3)
        sc = dataclasses.replace(sc, boxsize=100.0 * (n_part / 1.0e7) ** (1.0 / 3.0))
    L = sc.boxsize
    a = sc.a
    m_p = sc.mean_density * (a * L) ** 3 / n_part

    def rand(*shape):
        return torch.rand(*shape, generator=gen, device=dev, dtype=torch.float64)

    def randn(*shape):
        return torch.randn(*shape, generator=gen, device=dev, dtype=torch.float64)

    # ---- halo sizes: truncated power law, slope -1.9, then rescaled --------------
    # largest halo of the power law: 5 % of the box, at most `max_halo` particles (default 3e5: the BASELINE
    # configurations; bench.py --workload giant lifts it to 1e6)
    alpha = 1.9
    nmax = max(4 * nmin, min(3.0e5 if max_halo is None else float(max_halo), 0.05 * n_part))
    u = rand(n_halo)
    e = 1.0 - alpha
    ntot = (nmin**e + u * (nmax**e - nmin**e)) ** (1.0 / e)
    target = halo_fraction * n_part
    ntot = torch.clamp((ntot * (target / ntot.sum())).round().to(torch.int64), min=12)
    n_in_halos = int(ntot.sum().item())
    if n_in_halos > n_part:
        raise ValueError("halo_fraction too large for the requested halo count")
    n_bg = n_part - n_in_halos

    conc = 5.0 + 5.0 * rand(n_halo)
    centres = L * rand(n_halo, 3)
    tab_s, tab_m = _nfw_table(dev)
    m_c = torch.log1p(conc) - conc / (1.0 + conc)
    m_2c = torch.log1p(2 * conc) - 2 * conc / (1.0 + 2 * conc)
    # N200 = expected members inside R200c; defines M200 and hence R200c (physical -> comoving)
    n200 = ntot.to(torch.float64) * (m_c / m_2c)
    M200 = n200 * m_p
    r200 = (3.0 * M200 / (4.0 * math.pi * 200.0 * sc.critical_density)) ** (1.0 / 3.0) / a
    vbulk = 300.0 * randn(n_halo, 3)
    sigma = torch.sqrt(sc.G * M200 / (2.0 * r200 * a))

    # ---- halo particles ---------------------------------------------------------
    hid = torch.repeat_interleave(torch.arange(n_halo, device=dev), ntot)
    first = torch.zeros(n_in_halos, dtype=torch.bool, device=dev)
    starts = torch.cumsum(ntot, 0) - ntot
    first[starts] = True
    mu = rand(n_in_halos) * m_2c[hid]
    s = _interp(mu, tab_m, tab_s)
    r = s / conc[hid] * r200[hid]
    r[first] = 0.0
    cost = 2.0 * rand(n_in_halos) - 1.0
    sint = torch.sqrt((1.0 - cost) * (1.0 + cost))
    phi = 2.0 * math.pi * rand(n_in_halos)
    pos_h = torch.stack([r * sint * torch.cos(phi), r * sint * torch.sin(phi), r * cost], dim=1)
    pos_h = pos_h + centres[hid]
    # the central particle sits exactly on the halo centre
    pos_h[first] = centres
    vel_h = vbulk[hid] + sigma[hid, None] * randn(n_in_halos, 3)
    bound = r <= r200[hid]
    grnr_h = torch.where(bound, hid, torch.full_like(hid, -1)).to(torch.int32)
    fof_h = hid.to(torch.int64)

    # ---- background --------------------------------------------------------------
    pos_b = L * rand(n_bg, 3)
    vel_b = 100.0 * randn(n_bg, 3)
    grnr_b = torch.full((n_bg,), -1, dtype=torch.int32, device=dev)
    fof_b = torch.full((n_bg,), NO_FOF, dtype=torch.int64, device=dev)

    pos = torch.cat([pos_h, pos_b]) % L
    # halo centres are inside [0,L); the central particle must still equal the centre
    pos[:n_in_halos][first] = centres
    vel = torch.cat([vel_h, vel_b])
    grnr = torch.cat([grnr_h, grnr_b])
    fof = torch.cat([fof_h, fof_b])
    del pos_h, pos_b, vel_h, vel_b

    # ---- exact bound counts / search radii ------------------------------------------
    nr_bound = torch.zeros(n_halo, dtype=torch.int64, device=dev)
    nr_bound.scatter_add_(0, hid[bound], torch.ones(int(bound.sum()), dtype=torch.int64, device=dev))
    rmax = torch.zeros(n_halo, dtype=torch.float64, device=dev)
    rmax.scatter_reduce_(0, hid[bound], r[bound], reduce="amax", include_self=True)
    # r was computed before adding the centre; recompute the bound radii the way the
    # reference sees them (wrapped position minus centre) so that 1.01*rmax is safe
    search_radius = 1.01 * rmax + 1e-9 * L
    # clip to the largest physical radius a default calculation asks for is the harness'
    # job (halo_centres.py:245-251); here only a floor for degenerate 1-particle halos
    search_radius = torch.clamp(search_radius, min=1e-3 * r200.min().item())
    read_radius = torch.clamp(read_radius_factor * torch.maximum(search_radius, r200), max=0.45 * L)
    read_radius = torch.maximum(read_radius, search_radius)

    # ---- particle types ------------------------------------------------------------------
    if hydro:
        probs = torch.tensor(HYDRO_PROBS, dtype=torch.float64, device=dev)
        tsel = torch.multinomial(probs, n_part, replacement=True, generator=gen)
        ptype_of = torch.tensor(HYDRO_TYPES, device=dev)[tsel]
        # the central particle of each halo is DM so every type list keeps its meaning
        ptype_of[:n_in_halos][first] = 1
    else:
        ptype_of = torch.ones(n_part, dtype=torch.int64, device=dev)
    mass = torch.full((n_part,), m_p, dtype=torch.float64, device=dev)
    if hydro:
        # unequal masses per type exercise the weighted paths
        mass = mass * (0.5 + rand(n_part))
        if float32_masses:
            # the values a SWIFT snapshot holds (its Masses datasets are float32), kept in float64 arrays
            mass = mass.to(torch.float32).to(torch.float64)

    # ---- SWIFT-like storage order: by coarse top-level cell, random inside -----------------
    tc = torch.clamp((pos / (L / topcells)).floor().to(torch.int64), 0, topcells - 1)
    key = (tc[:, 0] * topcells + tc[:, 1]) * topcells + tc[:, 2]
    key = key * 4096 + torch.randint(0, 4096, (n_part,), generator=gen, device=dev)
    order = torch.argsort(key)
    del key, tc
    pos, vel, mass, grnr, fof, ptype_of = (
        pos[order], vel[order], mass[order], grnr[order], fof[order], ptype_of[order])  # fmt: skip

    data: Dict[int, Dict[str, torch.Tensor]] = {}
    for pt in HYDRO_TYPES if hydro else (1,):
        sel = ptype_of == pt if hydro else slice(None)
        data[pt] = {
            "Coordinates": pos[sel].contiguous(),
            "Velocities": vel[sel].contiguous(),
            "Masses": mass[sel].contiguous(),
            "GroupNr_bound": grnr[sel].contiguous(),
            "FOFGroupIDs": fof[sel].contiguous(),
        }

    is_central = torch.ones(n_halo, dtype=torch.int32, device=dev)
    if satellite_fraction > 0:
        is_central[rand(n_halo) < satellite_fraction] = 0
    horder = torch.argsort(nr_bound, descending=True, stable=True)
    halos = {
        "index": torch.arange(n_halo, device=dev, dtype=torch.int64)[horder],
        "cofp": centres[horder].contiguous(),
        "search_radius": search_radius[horder].contiguous(),
        "read_radius": read_radius[horder].contiguous(),
        "is_central": is_central[horder].contiguous(),
        "nr_bound_part": nr_bound[horder].contiguous(),
    }
    return sc, data, halos


def to_numpy(data, halos):
    d = {pt: {k: v.cpu().numpy() for k, v in dd.items()} for pt, dd in data.items()}
    h = {k: v.cpu().numpy() for k, v in halos.items()}
    return d, h
