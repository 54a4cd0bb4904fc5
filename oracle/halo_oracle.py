"""
oracle/halo_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU (NumPy/SciPy, fp64) restatement of SOAP's per-halo hot path:
particle_selection + property_calculation as driven by core/halo_tasks.py.
It exists to *check* the HIP engine (tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py).  Nothing in the product path (soap_amd/) may
import it.

Parity pinning: the arithmetic kernels in here (SO radius finder, half-mass
radius, v_max, angular momentum, dispersion matrix, mesh query) are pinned by
 (a) golden vectors generated in the build container by running the
     reference's own functions (tests/golden/make_golden.py, outputs committed
     under tests/golden/*.npz), and
 (b) the invariants the reference's own tests assert
     (tests/test_shared_mesh.py:76-124, tests/test_half_mass_radius.py:29-38,
      tests/test_SO_properties.py:434-446).
Each function cites the reference file:line it follows (paths relative to the
reference checkout, e.g. SOAP/core/halo_tasks.py).

Unit convention (the reference carries `unyt` units; this restatement is
unit-free, one fixed system, see DESIGN.md "Units"):
  lengths      comoving code length  (a * snap_length)
  velocities   code velocity         (snap_length / snap_time)
  masses       code mass
  densities    code mass / comoving code length^3
Physical radii (kpc apertures, SO-radius multiples) are converted to comoving
code length ONCE by `Units`, with a single explicit factor, so that membership
counts are reproducible bit-for-bit between oracle and engine.
"""

from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import numpy as np
from scipy.optimize import brentq

# SOAP/core/halo_tasks.py:14-20
SEARCH_RADIUS_FACTOR = 1.2
READ_RADIUS_FACTOR = 1.5

# SOAP/core/dataset_names.py:4
PTYPES_FOR_SO_MASSES = (0, 1, 4, 5)
# particle types the selections concatenate, in this order
# (SO_properties.py:3520-3550, aperture_properties.py:3995-4005)
PTYPES = (0, 1, 4, 5)


class SearchRadiusTooSmallError(Exception):
    """SOAP/particle_selection/halo_properties.py:4-5"""


# --------------------------------------------------------------------------
# units / snapshot scalars (what HaloProperty.__init__ copies from the cellgrid,
# SOAP/particle_selection/halo_properties.py:8-35)
# --------------------------------------------------------------------------
@dataclasses.dataclass
class Units:
    """All scalars the hot path needs, in the fixed unit system above."""

    a: float  # expansion factor
    boxsize: float  # comoving code length
    critical_density: float  # PHYSICAL code density (header value)
    mean_density: float  # PHYSICAL code density
    kpc: float  # one physical kpc in PHYSICAL code length (kpc_cm / U_L)
    G: float  # Newton's G in physical code units
    softening: Dict[int, float]  # per ptype, PHYSICAL code length
    nu_density: float = 0.0  # PHYSICAL code density of the nu background
    virBN98: float = 0.0
    # flow rates through spherical shells (SO_properties.py:3446-3451): Hubble rate in 1 / code time,
    # radiation and matter density parameters at this redshift
    H: float = 0.0
    Omega_g: float = 0.0
    Omega_m: float = 0.0

    # --- the single explicit conversions (DESIGN.md "Units") ---
    def kpc_to_comoving(self, r_kpc: float) -> float:
        return r_kpc * self.kpc / self.a

    def mpc_to_comoving(self, r_mpc: float) -> float:
        return r_mpc * (1000.0 * self.kpc) / self.a

    def density_to_comoving(self, rho_phys: float) -> float:
        return rho_phys * self.a**3

    def softening_comoving(self, ptype: int) -> float:
        return self.softening[ptype] / self.a

    @property
    def G_eff(self) -> float:
        # v^2 = G M / (a r_comoving)
        return self.G / self.a

    @property
    def nu_density_comoving(self) -> float:
        return self.nu_density * self.a**3


def flamingo_like_units(boxsize: float = 100.0) -> Units:
    """
    The concrete consistent scalar set of the reference's dummy snapshot
    (tests/dummy_halo_generator.py:40-126, :416-496): FLAMINGO header at z=0.3.
    """
    a = 0.76923077
    U_L = 3.08567758e24
    U_M = 1.98841e43
    U_t = 3.08567758e19
    rho_crit = 17.58736923
    Omega_m = 0.304611
    Omega_k = 2.5212783e-09
    Omega_lambda = 0.693922
    G_cgs = 6.6743e-08
    kpc_cm = 3.08567758e21
    soft = min(0.0446 * a, 0.0114)
    bnx = -(Omega_k / a**2 + Omega_lambda) / (
        Omega_k / a**2 + Omega_m / a**3 + Omega_lambda
    )
    virBN98 = 18.0 * np.pi**2 + 82.0 * bnx - 39.0 * bnx**2
    return Units(
        a=a,
        boxsize=boxsize,
        critical_density=rho_crit,
        mean_density=rho_crit * Omega_m,
        kpc=kpc_cm / U_L,
        G=G_cgs * U_M * U_t**2 / U_L**3,
        softening={0: soft, 1: soft, 4: soft, 5: soft},
        nu_density=0.0,
        virBN98=float(virBN98),
        H=79.60499176,
        Omega_g=5.33243487e-05,
        Omega_m=Omega_m,
    )


# --------------------------------------------------------------------------
# L0 kernels
# --------------------------------------------------------------------------
def _qpow(x, p):
    """
    `x**p` for a unyt QUANTITY.  In the reference these are 0-d arrays, so the power goes
    through NumPy's array loop (a SIMD pow on AVX-512/AVX2 hosts) and not through the scalar
    libm pow that a bare np.float64 would use; the two differ in the last bit for a few per
    cent of arguments.  Mirrored here so that golden vectors generated from the reference
    match bit-for-bit on the same host.
    """
    return np.power(np.asarray(x, dtype=np.float64), p)[()]


def cumulative_mass_intersection(r, rho_dim, slope_dim):
    """SO_properties.py:50-77"""
    return 4.0 * np.pi / 3.0 * rho_dim * r**3 - slope_dim * r + slope_dim - 1.0


def find_SO_radius_and_mass(
    ordered_radius, density, cumulative_mass, reference_density, mpc20=np.inf
):
    """
    SO_properties.py:80-217.  `mpc20` is 20 Mpc in the radius array's unit
    (the RuntimeError branch, :150-153, :190-193).
    Returns (SO_r, SO_mass, SO_volume).
    """
    above_mask = density > reference_density
    if above_mask[0]:
        below_mask = ~above_mask
        i = int(np.argmax(below_mask))
        if i == 0:
            if ordered_radius[-1] > mpc20:
                raise RuntimeError(
                    "Cannot find SO radius, but search radius is already larger than 20 Mpc!"
                )
            raise SearchRadiusTooSmallError("SO radius multiple estimate was too small!")
    else:
        ipos = 0
        while ipos < len(cumulative_mass) and cumulative_mass[ipos] < 0.0:
            ipos += 1
        if ipos == len(cumulative_mass):
            raise RuntimeError("Should never happen!")
        SO_r = np.sqrt(
            0.75 * cumulative_mass[ipos] / (np.pi * ordered_radius[ipos] * reference_density)
        )
        SO_mass = cumulative_mass[ipos] * SO_r / ordered_radius[ipos]
        return SO_r, SO_mass, 4.0 * np.pi / 3.0 * _qpow(SO_r, 3)

    r1 = ordered_radius[i - 1]
    r2 = ordered_radius[i]
    M1 = cumulative_mass[i - 1]
    M2 = cumulative_mass[i]
    while r1 == r2 or (above_mask[i - 1] == above_mask[i]):
        i += 1
        if i >= len(density):
            if ordered_radius[-1] > mpc20:
                raise RuntimeError(
                    "Cannot find SO radius, but search radius is already larger than 20 Mpc!"
                )
            raise SearchRadiusTooSmallError("SO radius multiple estimate was too small!")
        r1 = r2
        r2 = ordered_radius[i]
        M1 = M2
        M2 = cumulative_mass[i]

    rho_dim = reference_density * _qpow(r1, 3) / M1
    slope_dim = (M2 - M1) / (r2 - r1) * (r1 / M1)
    SO_r = r1 * brentq(cumulative_mass_intersection, 1.0, r2 / r1, args=(rho_dim, slope_dim))
    SO_volume = 4.0 / 3.0 * np.pi * _qpow(SO_r, 3)
    SO_mass = SO_volume * reference_density
    return SO_r, SO_mass, SO_volume


def get_half_weight_radius(radius, weights, total_weight):
    """half_mass_radius.py:16-97"""
    if total_weight == 0.0 or len(weights) < 1:
        return 0.0
    target_weight = 0.5 * total_weight
    isort = np.argsort(radius)
    sorted_radius = radius[isort]
    cumulative_weights = weights[isort].cumsum(dtype=np.float64)
    if cumulative_weights[-1] < 0.998 * total_weight:
        raise RuntimeError("Weights sum up to less than the given total weight value")
    ihalf = int(np.argmax(cumulative_weights >= target_weight))
    if ihalf == 0:
        rmin = 0.0
        WeightMin = 0.0
    else:
        rmin = sorted_radius[ihalf - 1]
        WeightMin = cumulative_weights[ihalf - 1]
    rmax = sorted_radius[ihalf]
    WeightMax = cumulative_weights[ihalf]
    if WeightMin == WeightMax:
        half_weight_radius = 0.5 * (rmin + rmax)
    else:
        half_weight_radius = rmin + (target_weight - WeightMin) / (WeightMax - WeightMin) * (
            rmax - rmin
        )
    if half_weight_radius > sorted_radius[-1]:
        raise RuntimeError("Half weight radius larger than input radii")
    return float(half_weight_radius)


def get_vmax(mass, radius, G, nskip=0):
    """kinematic_properties.py:555-593.  Returns (r_vmax, vmax)."""
    isort = np.argsort(radius)
    ordered_radius = radius[isort]
    cumulative_mass = mass[isort].cumsum()
    nskip = max(nskip, int(np.argmin(np.isclose(ordered_radius, 0.0))))
    ordered_radius = ordered_radius[nskip:]
    if len(ordered_radius) == 0 or ordered_radius[0] == 0:
        return 0.0, 0.0
    cumulative_mass = cumulative_mass[nskip:]
    v_over_G = cumulative_mass / ordered_radius
    imax = int(np.argmax(v_over_G))
    return float(ordered_radius[imax]), float(np.sqrt(v_over_G[imax] * G))


def get_angular_momentum(mass, position, velocity, ref_velocity=None):
    """kinematic_properties.py:222-263 (also :325-356 for the mass-weighted path)"""
    vrel = velocity if ref_velocity is None else velocity - ref_velocity[None, :]
    return (mass[:, None] * np.cross(position, vrel)).sum(axis=0)


def get_velocity_dispersion_matrix(mass_fraction, velocity, ref_velocity):
    """
    kinematic_properties.py:91-127.  The reference accumulates into a float32
    buffer (:117); we return (fp64 pre-cast values, float32 stored values).
    """
    vrel = velocity - ref_velocity[None, :]
    r64 = np.array(
        [
            (mass_fraction * vrel[:, 0] * vrel[:, 0]).sum(),
            (mass_fraction * vrel[:, 1] * vrel[:, 1]).sum(),
            (mass_fraction * vrel[:, 2] * vrel[:, 2]).sum(),
            (mass_fraction * vrel[:, 0] * vrel[:, 1]).sum(),
            (mass_fraction * vrel[:, 0] * vrel[:, 2]).sum(),
            (mass_fraction * vrel[:, 1] * vrel[:, 2]).sum(),
        ]
    )
    r32 = np.zeros(6, dtype=np.float32)
    r32 += r64
    return r64, r32


def concentration_from_R1(R1):
    """SO_properties.py:2725-2736; result is float32 in the reference."""
    polynomial = [-79.71, -222.46, -250.14, -140.17, -43.59, -5.07]
    c = 0
    for i, b in enumerate(polynomial[::-1]):
        c += b * np.log10(R1) ** i
    c = max(min(c, 3), 0)
    return float(10**c)


# --------------------------------------------------------------------------
# mesh (SOAP/core/shared_mesh.py)
# --------------------------------------------------------------------------
class Mesh:
    """shared_mesh.py:11-114 (single process: no MPI reductions)."""

    def __init__(self, pos: np.ndarray, resolution: int):
        self.empty = pos.shape[0] == 0
        if self.empty:
            return
        self.pos_min = np.amin(pos, axis=0)
        self.pos_max = np.amax(pos, axis=0)
        for i in range(3):
            if self.pos_min[i] == self.pos_max[i]:
                self.pos_max[i] = self.pos_min[i] + 1.0
        self.resolution = int(resolution)
        nr_cells = self.resolution**3
        self.cell_size = (self.pos_max - self.pos_min) / self.resolution
        cell_idx = np.floor((pos - self.pos_min[None, :]) / self.cell_size[None, :]).astype(
            np.int32
        )
        cell_idx = np.clip(cell_idx, 0, self.resolution - 1)
        cell_idx = (
            cell_idx[:, 0]
            + self.resolution * cell_idx[:, 1]
            + (self.resolution**2) * cell_idx[:, 2]
        )
        self.cell_count = np.bincount(cell_idx, minlength=nr_cells)
        self.cell_offset = np.zeros_like(self.cell_count)
        if nr_cells > 1:
            self.cell_offset[1:] = np.cumsum(self.cell_count[:-1])
        # the reference uses an MPI parallel sort (shared_mesh.py:105); ordering
        # within a cell only affects summation order -> use a stable sort.
        self.sort_idx = np.argsort(cell_idx, kind="stable")

    def cell_coords(self, centre, radius, boxsize):
        """shared_mesh.py:146-179 -> sorted lists of cell coordinates per dimension."""
        coords = [set() for _ in range(3)]
        for dim in (0, 1, 2):
            min_copy_nr = 0
            while centre[dim] + (min_copy_nr - 1) * boxsize + radius >= self.pos_min[dim]:
                min_copy_nr -= 1
            max_copy_nr = 0
            while centre[dim] + (max_copy_nr + 1) * boxsize - radius <= self.pos_max[dim]:
                max_copy_nr += 1
            for copy_nr in range(min_copy_nr, max_copy_nr + 1, 1):
                min_coord = max(self.pos_min[dim], centre[dim] + copy_nr * boxsize - radius)
                min_idx = int(np.floor((min_coord - self.pos_min[dim]) / self.cell_size[dim]))
                max_coord = min(self.pos_max[dim], centre[dim] + copy_nr * boxsize + radius)
                max_idx = int(np.floor((max_coord - self.pos_min[dim]) / self.cell_size[dim]))
                for cell_nr in range(min_idx, max_idx + 1):
                    if cell_nr >= 0 and cell_nr < self.resolution:
                        coords[dim].add(cell_nr)
        # Python set iteration order is an implementation detail; we fix it to
        # ascending (order only affects summation rounding).
        return [sorted(c) for c in coords]

    def query_radius_periodic(self, centre, radius, pos, boxsize, return_ncand=False):
        """shared_mesh.py:122-200"""
        if self.empty:
            out = np.ndarray(0, dtype=np.int64)
            return (out, 0) if return_ncand else out
        cc = self.cell_coords(centre, radius, boxsize)
        idx = []
        ncand = 0
        res = self.resolution
        r2max = radius * radius
        for k in cc[2]:
            for j in cc[1]:
                # the i-cells of one (j,k) row are consecutive in memory; evaluate the
                # row cell by cell exactly as the reference does.
                for i in cc[0]:
                    cell_nr = i + res * j + (res**2) * k
                    start = self.cell_offset[cell_nr]
                    count = self.cell_count[cell_nr]
                    if count > 0:
                        ncand += int(count)
                        idx_in_cell = self.sort_idx[start : start + count]
                        dr = pos[idx_in_cell, :] - centre[None, :]
                        dr[dr > 0.5 * boxsize] -= boxsize
                        dr[dr < -0.5 * boxsize] += boxsize
                        r2 = np.sum(dr**2, axis=1)
                        keep = r2 <= r2max
                        if np.sum(keep) > 0:
                            idx.append(idx_in_cell[keep])
        out = np.concatenate(idx) if len(idx) > 0 else np.ndarray(0, dtype=np.int64)
        return (out, ncand) if return_ncand else out


def mesh_resolution(nr_parts: int) -> int:
    """SOAP/core/chunk_tasks.py:299-302"""
    resolution = int((nr_parts / 1000) ** (1.0 / 3.0))
    return min(max(resolution, 1), 256)


# --------------------------------------------------------------------------
# category filter (SOAP/core/category_filter.py:69-106)
# --------------------------------------------------------------------------
DEFAULT_FILTERS = {
    "general": dict(limit=100, properties=("Ndm", "Ngas", "Nstar", "Nbh")),
    "dm": dict(limit=100, properties=("Ndm",)),
    "gas": dict(limit=100, properties=("Ngas",)),
    "star": dict(limit=100, properties=("Nstar",)),
    "baryon": dict(limit=100, properties=("Ngas", "Nstar")),
}


def get_do_calculation(filters, counts, dmo):
    """counts: dict Ndm/Ngas/Nstar/Nbh of the bound subhalo."""
    do_calculation = {"basic": True}
    c = dict(counts)
    if dmo:
        c["Ngas"] = 0
        c["Nstar"] = 0
        c["Nbh"] = 0
    for name, info in filters.items():
        v = 0
        for p in info["properties"]:
            v += c[p]
        do_calculation[name] = v >= info["limit"]
    return do_calculation


# --------------------------------------------------------------------------
# per-halo particle container helpers
# --------------------------------------------------------------------------
def _concat_types(data, fields):
    out = {f: [] for f in fields}
    types = []
    for ptype in PTYPES:
        if ptype not in data:
            continue
        d = data[ptype]
        n = d["Coordinates"].shape[0]
        for f in fields:
            out[f].append(d[f])
        types.append(ptype * np.ones(n, dtype=np.int32))
    res = {f: np.concatenate(v) if len(v) else np.zeros(0) for f, v in out.items()}
    res["types"] = np.concatenate(types) if len(types) else np.zeros(0, dtype=np.int32)
    return res


# Output bookkeeping: each calculation fills halo_result[f"{group}/{key}"] with a
# `Value(raw, stored)`; raw = fp64 pre-cast value in engine units, stored = what
# the reference would put in its zero-initialised table-dtype array
# (SO_properties.py:3600-3620, :3682-3699): float32/float64/uint32.
@dataclasses.dataclass
class Value:
    raw: np.ndarray
    stored: np.ndarray


# key -> (shape, dtype).  Mirrors the rows of SOAP/property_table.py listed in
# SURVEY.md section 8(a) a22.  Units/a-exponents live in soap_amd/property_table.py.
SCHEMA = {
    "r": (1, np.float32),
    "Mtot": (1, np.float32),
    "Mdm": (1, np.float32),
    "Mgas": (1, np.float32),
    "Mstar": (1, np.float32),
    "Mbh_dynamical": (1, np.float32),
    "Ndm": (1, np.uint32),
    "Ngas": (1, np.uint32),
    "Nstar": (1, np.uint32),
    "Nbh": (1, np.uint32),
    "com": (3, np.float64),
    "vcom": (3, np.float32),
    "Ldm": (3, np.float32),
    "veldisp_matrix_dm": (6, np.float32),
    "proj_veldisp_dm": (1, np.float32),
    "HalfMassRadiusDM": (1, np.float32),
    "HalfMassRadiusTot": (1, np.float32),
    "EncloseRadius": (1, np.float32),
    "Vmax_soft": (1, np.float32),
    "R_vmax_soft": (1, np.float32),
    "Vmax_unsoft": (1, np.float32),
    "R_vmax_unsoft": (1, np.float32),
    "spin_parameter": (1, np.float32),
    "concentration_soft": (1, np.float32),
    "concentration_unsoft": (1, np.float32),
    "Mfrac_satellites": (1, np.float32),
    "Mfrac_external": (1, np.float32),
    # gas / star / baryon kinematics (property_table.py rows of SURVEY.md 8(a) a22, hydro part)
    "com_gas": (3, np.float64),
    "com_star": (3, np.float64),
    "vcom_gas": (3, np.float32),
    "vcom_star": (3, np.float32),
    "Lgas": (3, np.float32),
    "Lstar": (3, np.float32),
    "Lbaryons": (3, np.float32),
    "veldisp_matrix_gas": (6, np.float32),
    "veldisp_matrix_star": (6, np.float32),
    "proj_veldisp_gas": (1, np.float32),
    "proj_veldisp_star": (1, np.float32),
    "HalfMassRadiusGas": (1, np.float32),
    "HalfMassRadiusStar": (1, np.float32),
    "HalfMassRadiusBaryon": (1, np.float32),
    # ordered-rotation diagnostics (property_table.py:819-841, :3794-3835)
    "kappa_corot_gas": (1, np.float32),
    "kappa_corot_star": (1, np.float32),
    "kappa_corot_baryons": (1, np.float32),
    "DtoTgas": (1, np.float32),
    "DtoTstar": (1, np.float32),
    # 2D inertia tensors of projected apertures (property_table.py:2150-2280): (1,1), (2,2), (1,2)
    **{f"Projected{c}InertiaTensor{s_}": (3, np.float32) for c in ("Total", "Gas", "Stellar")
       for s_ in ("", "Reduced", "Noniterative", "ReducedNoniterative")},
    # mass fraction of the bound stars born in another subhalo: dark-matter-only runs carry the (zero) column
    # (aperture_properties.py:764-780, subhalo_properties.py; `Nstar == 0` gives None); with stars it needs
    # PartType4/BirthHaloCatalogueIndex, which is outside this path
    "ExSituFraction": (1, np.float32),
    # inertia tensors of the total and the dark matter mass (property_table.py:2724-2790 and the DarkMatter rows):
    # xx yy zz xy xz yz
    **{k: (6, np.float32) for k in (
        "TotalInertiaTensor", "TotalInertiaTensorReduced", "TotalInertiaTensorNoniterative",
        "TotalInertiaTensorReducedNoniterative", "DarkMatterInertiaTensor", "DarkMatterInertiaTensorReduced",
        "DarkMatterInertiaTensorNoniterative", "DarkMatterInertiaTensorReducedNoniterative",
        "GasInertiaTensor", "GasInertiaTensorReduced", "GasInertiaTensorNoniterative",
        "GasInertiaTensorReducedNoniterative", "StellarInertiaTensor", "StellarInertiaTensorReduced",
        "StellarInertiaTensorNoniterative", "StellarInertiaTensorReducedNoniterative")},
    # neutrino particles inside an SO (property_table.py: NumberOfNeutrinoParticles, RawNeutrinoMass,
    # NoiseSuppressedNeutrinoMass)
    "Nnu": (1, np.uint32),
    "Mnu": (1, np.float32),
    "MnuNS": (1, np.float32),
    # inflow at 0.1 / 0.3 / 1 R_SO, outflow at 0.1 / 0.3 / 1 R_SO (property_table.py:3327-3342)
    "DarkMatterMassFlowRate": (6, np.float32),
}


INERTIA_KEYS = (
    "TotalInertiaTensor", "TotalInertiaTensorReduced",
    "TotalInertiaTensorNoniterative", "TotalInertiaTensorReducedNoniterative",
    "DarkMatterInertiaTensor", "DarkMatterInertiaTensorReduced",
    "DarkMatterInertiaTensorNoniterative", "DarkMatterInertiaTensorReducedNoniterative",
    "GasInertiaTensor", "GasInertiaTensorReduced",
    "GasInertiaTensorNoniterative", "GasInertiaTensorReducedNoniterative",
    "StellarInertiaTensor", "StellarInertiaTensorReduced",
    "StellarInertiaTensorNoniterative", "StellarInertiaTensorReducedNoniterative",
)


def get_weighted_inertia_tensor(particle_weights, particle_positions, sphere_radius, search_radius=None,
                                reduced=False, max_iterations=20, min_particles=20):  # fmt: skip
    """
    property_calculation/inertia_tensors.py:19-132.  I_ij = sum w x_i x_j / sum w over the particles inside an
    ellipsoid that starts as the sphere of `sphere_radius` and is re-shaped from the eigenvectors / eigenvalue
    ratios of the previous tensor (same volume), until q = sqrt(b/a) changes by less than 1e-4 or
    `max_iterations` tensors were computed.  `reduced` divides every particle's term by |x|^2 (particles at
    the centre, np.isclose(|x|^2, 0) in the position unit, are dropped first).  Returns the six independent
    components (xx, yy, zz, xy, xz, yz) of the LAST tensor, or None with fewer than `min_particles`
    particles (in total, or inside the initial sphere).  The reference works in kpc; a uniform rescaling of
    the lengths changes neither the memberships nor the eigenvectors, so this works in the caller's unit.
    """
    if particle_weights.shape[0] < min_particles:
        return None
    if reduced:
        norm = np.linalg.norm(particle_positions, axis=1) ** 2
        mask = np.logical_not(np.isclose(norm, 0))
        norm = norm[mask]
        particle_weights = particle_weights[mask]
        particle_positions = particle_positions[mask]
    tol = 0.0001
    q = 1000
    R = sphere_radius
    eig_val = [1, 1, 1]
    eig_vec = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    tensor = None
    for i_iter in range(max_iterations):
        old_q = q
        with np.errstate(all="ignore"):
            q = np.sqrt(eig_val[1] / eig_val[2])
            s = np.sqrt(eig_val[0] / eig_val[2])
            p = np.sqrt(eig_val[0] / eig_val[1])
            if abs((old_q - q) / q) < tol:
                break
            axis = R * np.array([1 * np.cbrt(s * p), 1 * np.cbrt(q / p), 1 / np.cbrt(q * s)])
            p = np.dot(particle_positions, eig_vec) / axis
            r = np.linalg.norm(p, axis=1)
            if (i_iter == 0) and (np.sum(r <= 1) < min_particles):
                return None
            weight = particle_weights / np.sum(particle_weights[r <= 1])
            weight[r > 1] = 0
            if (search_radius is not None) and (np.max(R) > search_radius):
                raise SearchRadiusTooSmallError("Inertia tensor required more particles")
            tensor = weight[:, None, None] * particle_positions[:, :, None] * particle_positions[:, None, :]
            if reduced:
                tensor /= norm[:, None, None]
            tensor = tensor.sum(axis=0)
            eig_val, eig_vec = np.linalg.eigh(tensor)
            eig_val = np.abs(eig_val)
        if q == 0:
            tensor.fill(0)
            break
    return np.concatenate([np.diag(tensor), tensor[np.triu_indices(3, 1)]])


# the parameter file can switch properties off (parameter_file.py:85-154); the only switch this restatement
# needs is the one for the inertia tensors, whose iterations dominate the run time when they are on
ENABLE_INERTIA = True


PROJECTED_INERTIA_KEYS = tuple(
    f"Projected{comp}InertiaTensor{suffix}"
    for comp in ("Total", "Gas", "Stellar")
    for suffix in ("", "Reduced", "Noniterative", "ReducedNoniterative")
)


def get_weighted_projected_inertia_tensor(particle_weights, particle_positions, axis, radius, reduced=False,
                                          max_iterations=20, min_particles=20):  # fmt: skip
    """
    property_calculation/inertia_tensors.py:226-343: the same iteration in projection along `axis` (0: (y, z),
    1: (z, x), 2: (x, y)), with an ellipse of the area of the circle of `radius`; eigenvalues are used as they
    come (no absolute value), there is no search-radius condition.  Returns (I_11, I_22, I_12) or None.
    """
    if particle_weights.shape[0] < min_particles:
        return None
    projected_position = np.zeros((particle_positions.shape[0], 2), dtype=particle_positions.dtype)
    if axis == 0:
        projected_position[:, 0] = particle_positions[:, 1]
        projected_position[:, 1] = particle_positions[:, 2]
    elif axis == 1:
        projected_position[:, 0] = particle_positions[:, 2]
        projected_position[:, 1] = particle_positions[:, 0]
    elif axis == 2:
        projected_position[:, 0] = particle_positions[:, 0]
        projected_position[:, 1] = particle_positions[:, 1]
    else:
        raise AttributeError(f"Invalid axis: {axis}!")
    if reduced:
        norm = np.linalg.norm(projected_position, axis=1) ** 2
        mask = np.logical_not(np.isclose(norm, 0))
        norm = norm[mask]
        particle_weights = particle_weights[mask]
        projected_position = projected_position[mask]
    tol = 0.0001
    q = 1000
    R = radius
    eig_val = [1, 1]
    eig_vec = np.array([[1, 0], [0, 1]])
    tensor = None
    for i_iter in range(max_iterations):
        old_q = q
        with np.errstate(all="ignore"):
            q = np.sqrt(eig_val[0] / eig_val[1])
            if abs((old_q - q) / q) < tol:
                break
            ax = R * np.array([1 * np.sqrt(q), 1 / np.sqrt(q)])
            p = np.dot(projected_position, eig_vec) / ax
            r = np.linalg.norm(p, axis=1)
            if (i_iter == 0) and (np.sum(r <= 1) < min_particles):
                return None
            weight = particle_weights / np.sum(particle_weights[r <= 1])
            weight[r > 1] = 0
            tensor = weight[:, None, None] * projected_position[:, :, None] * projected_position[:, None, :]
            if reduced:
                tensor /= norm[:, None, None]
            tensor = tensor.sum(axis=0)
            eig_val, eig_vec = np.linalg.eigh(tensor)
        if q == 0:
            tensor.fill(0)
            break
    return np.concatenate([np.diag(tensor), [tensor[(0, 1)]]])


def inertia_tensors(out, mass, position, types, radii, search_radius, has, n_inner=None):
    """
    The sixteen mass-weighted tensors of a selection (SO_properties.py:620-672, :798-857, :915-972, :1160-1218;
    subhalo_properties.py:1075-1125, :1206-1256, :1302-1352, :1730-1783): total mass, dark matter, gas and
    stars (`radii` and `has` in that order), iterative and within the fixed sphere (`max_iterations=1`), plain
    and reduced.  `mass / position / types`: every particle the iterative
    version may use (SO: members first, then the surrounding particles of the search sphere; bound subhalo:
    the bound particles); the fixed-sphere version is handed the same arrays, since particles outside the
    sphere get weight zero.  n_inner: the fixed-sphere versions only see the
    first n_inner particles (SO: the members, SO_properties.py:658-659, :957-958).
    """
    if not ENABLE_INERTIA:
        return
    n_inner = len(mass) if n_inner is None else n_inner
    inner = np.arange(len(mass)) < n_inner
    comps = (("Total", np.ones(len(mass), dtype=bool)), ("DarkMatter", types == 1), ("Gas", types == 0), ("Stellar", types == 4))
    for (name, sel), R, ok in zip(comps, radii, has):
        if not ok:
            continue
        m, x = mass[sel], position[sel]
        mi, xi = mass[sel & inner], position[sel & inner]
        for suffix, kw in (("", {}), ("Reduced", {"reduced": True})):
            out[f"{name}InertiaTensor{suffix}"] = get_weighted_inertia_tensor(m, x, R, search_radius=search_radius, **kw)
            out[f"{name}InertiaTensor{suffix}Noniterative"] = get_weighted_inertia_tensor(mi, xi, R, max_iterations=1, **kw)


def angular_momentum_and_kappa_corot(mass, position, velocity, ref_velocity, do_counterrot_mass=False):
    """
    get_angular_momentum_and_kappa_corot_weighted without weights (kinematic_properties.py:266-425, called
    through get_angular_momentum_and_kappa_corot_mass_weighted :428-456): total angular momentum about
    `ref_velocity`, kappa_corot (Correa et al. 2017: kinetic energy in ordered co-rotation over the total
    kinetic energy) and optionally the mass on counter-rotating orbits.  The reference keeps kappa and the
    counter-rotating mass in float32 buffers (:360-381); fp64 values are returned here, the float32 cast happens
    at the store.
    """
    prel = position
    vrel = velocity - ref_velocity[None, :]
    Lpart = mass[:, None] * np.cross(prel, vrel)
    Ltot = 1 * (np.ones(1)[:, None] * Lpart).sum(axis=0)
    Lnrm = np.linalg.norm(Ltot)
    kappa_corot = 0.0
    M_counterrot = 0.0
    if Lnrm > 0.0:
        K = 0.5 * (mass[:, None] * vrel**2).sum()
        if K > 0.0 or do_counterrot_mass:
            Ldir = Ltot / Lnrm
            Li = (Lpart * Ldir[None, :]).sum(axis=1)
        if K > 0.0:
            r2 = prel[:, 0] ** 2 + prel[:, 1] ** 2 + prel[:, 2] ** 2
            rdotL = (prel * Ldir[None, :]).sum(axis=1)
            Ri2 = r2 - rdotL**2
            mask = Ri2 == 0.0
            Ri2[mask] = 1.0
            Krot = 0.5 * (Li**2 / (mass * Ri2))
            Kcorot = Krot[(~mask) & (Li > 0.0)].sum()
            kappa_corot += Kcorot / K
        if do_counterrot_mass:
            M_counterrot += mass[Li < 0.0].sum()
    return Ltot, kappa_corot, M_counterrot


def baryon_kinematics(out, mass, position, velocity, types, radius, centre, boxsize, baryon_mass_from_parts=True):
    """
    Gas, star and baryon (gas + star) kinematics of a selection, written into `out`:
    centres of mass and velocities, angular momenta about the component's own centre-of-mass velocity,
    velocity dispersion tensors, half-mass radii.  One restatement for the four particle-data classes:
    SO_properties.py:716-783 (gas), :995-1087 (stars), :1225-1276 (baryons);
    aperture_properties.py:1129-1217, :1283-1360, :1466-1474, :1645-1706, :3469-3548;
    subhalo_properties.py (same bodies on the bound set).  Positions are relative to the halo centre.
    Keys that the reference leaves as None (component mass zero) are simply not set.
    """
    gas, star = types == 0, types == 4
    bar = gas | star
    for key, sel in (("gas", gas), ("star", star)):
        M = mass[sel].sum()
        if M != 0:
            f = mass[sel] / M
            out[f"com_{key}"] = ((f[:, None] * position[sel]).sum(axis=0) + centre) % boxsize
            vc = (f[:, None] * velocity[sel]).sum(axis=0)
            out[f"vcom_{key}"] = vc
            out[f"L{key}"] = get_angular_momentum(mass[sel], position[sel], velocity[sel], vc)
            out[f"veldisp_matrix_{key}"] = get_velocity_dispersion_matrix(f, velocity[sel], vc)[0]
            # compute_Lgas_props / compute_Lstar_props (aperture_properties.py:1147-1206, :1303-1422;
            # SO_properties.py:753-790, :1032-1111; subhalo_properties.py:1150-1200, :1570-1690)
            _, kappa, Mcr = angular_momentum_and_kappa_corot(mass[sel], position[sel], velocity[sel], vc, True)
            out[f"kappa_corot_{key}"] = kappa
            out[f"DtoT{key}"] = 1.0 - 2.0 * Mcr / M
    # the baryon mass: Mgas + Mstar in the aperture and subhalo classes (aperture_properties.py:535-539, subhalo_properties.py:1861-1865,
    # :2330-2334), one sum over the gas and star particles in the SO class (SO_properties.py:1240-1244) -- the same number in float64
    # once stored, not the same float32 number when the snapshot's masses are float32
    Mb = (mass[gas].sum() + mass[star].sum()) if baryon_mass_from_parts else mass[bar].sum()
    if Mb != 0:
        fb = mass[bar] / Mb
        vcb = (fb[:, None] * velocity[bar]).sum(axis=0)
        out["Lbaryons"] = get_angular_momentum(mass[bar], position[bar], velocity[bar], vcb)
        out["kappa_corot_baryons"] = angular_momentum_and_kappa_corot(mass[bar], position[bar], velocity[bar], vcb)[1]
    out["HalfMassRadiusGas"] = get_half_weight_radius(radius[gas], mass[gas], mass[gas].sum())
    out["HalfMassRadiusStar"] = get_half_weight_radius(radius[star], mass[star], mass[star].sum())
    out["HalfMassRadiusBaryon"] = get_half_weight_radius(radius[bar], mass[bar], Mb)


def _store(result, group, key, val):
    shape, dtype = SCHEMA[key]
    if val is None:
        raw = np.zeros(shape, dtype=np.float64)
    else:
        raw = np.atleast_1d(np.asarray(val, dtype=np.float64)).copy()
        assert raw.shape == (shape,), f"{key}: wrong shape {raw.shape}"
        assert np.max(np.abs(raw)) < float("inf")
    stored = np.zeros(shape, dtype=dtype)
    if np.issubdtype(dtype, np.integer):
        stored[:] = raw.astype(dtype)
    else:
        stored += raw  # zero-initialised table-dtype array += value (cast once)
    result[f"{group}/{key}"] = Value(raw, stored)


# --------------------------------------------------------------------------
# BoundSubhalo  (subhalo_properties.py)
# --------------------------------------------------------------------------
class SubhaloProperties:
    """subhalo_properties.py:128-164, :206-376, :798-846, :982-1073, :1272-1366,
    :2265-2343, :2587-2740 -- DMO/basic property subset."""

    name = "bound_subhalo"
    group_name = "BoundSubhalo"
    halo_filter = "basic"
    physical_radius_mpc = 0.0
    mean_density_multiple = None
    critical_density_multiple = None
    property_list = [
        "Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar", "Nbh",
        "com", "vcom", "Ldm", "Vmax_unsoft", "Vmax_soft", "R_vmax_unsoft", "spin_parameter",
        "HalfMassRadiusTot", "HalfMassRadiusDM", "veldisp_matrix_dm", "EncloseRadius",
        "com_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas", "veldisp_matrix_star",
        "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
        "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar",
        *INERTIA_KEYS,
    ]  # fmt: skip

    def __init__(self, units: Units, filters=DEFAULT_FILTERS, dmo=True):
        self.units = units
        self.filters = filters
        self.dmo = dmo

    def calculate(self, input_halo, search_radius, data, halo_result):
        u = self.units
        centre = input_halo["cofp"]
        index = input_halo["index"]
        mass, position, radius, velocity, types, softening = [], [], [], [], [], []
        for ptype in PTYPES:
            if ptype not in data:
                continue
            d = data[ptype]
            in_halo = d["GroupNr_bound"] == index
            mass.append(d["Masses"][in_halo])
            pos = d["Coordinates"][in_halo, :] - centre[None, :]
            position.append(pos)
            r = np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2 + pos[:, 2] ** 2)
            radius.append(r)
            velocity.append(d["Velocities"][in_halo, :])
            types.append(ptype * np.ones(r.shape, dtype=np.int32))
            softening.append(np.ones(r.shape, dtype=np.float64) * u.softening_comoving(ptype))
        mass = np.concatenate(mass)
        position = np.concatenate(position)
        radius = np.concatenate(radius)
        velocity = np.concatenate(velocity)
        types = np.concatenate(types)
        softening = np.concatenate(softening)

        N = {t: int((types == t).sum()) for t in PTYPES}
        counts = {"Ngas": N[0], "Ndm": N[1], "Nstar": N[4], "Nbh": N[5]}
        Ntot = N[0] + N[1] + N[4] + N[5]
        Nexpected = int(input_halo["nr_bound_part"])
        if Ntot < Nexpected:
            raise SearchRadiusTooSmallError(
                "Search radius does not contain expected number of particles!"
            )
        elif Ntot > Nexpected:
            raise RuntimeError(f"Found more particles than expected for halo {index}")

        g = self.group_name
        out = {}
        Mtot = mass.sum()
        out["Mtot"] = Mtot
        M_t = {t: mass[types == t].sum() for t in PTYPES}
        out["Mgas"], out["Mdm"], out["Mstar"], out["Mbh_dynamical"] = M_t[0], M_t[1], M_t[4], M_t[5]
        out["Ngas"], out["Ndm"], out["Nstar"], out["Nbh"] = N[0], N[1], N[4], N[5]
        vcom = None
        if Mtot != 0:
            frac = mass / Mtot
            out["com"] = ((frac[:, None] * position).sum(axis=0) + centre) % u.boxsize
            vcom = (frac[:, None] * velocity).sum(axis=0)
            out["vcom"] = vcom
        dm = types == 1
        if M_t[1] != 0:
            fdm = mass[dm] / M_t[1]
            vcom_dm = (fdm[:, None] * velocity[dm]).sum(axis=0)
            out["Ldm"] = get_angular_momentum(mass[dm], position[dm], velocity[dm], vcom_dm)
            out["veldisp_matrix_dm"] = get_velocity_dispersion_matrix(fdm, velocity[dm], vcom_dm)[0]
        if Mtot != 0:
            r_u, v_u = get_vmax(mass, radius, u.G_eff, nskip=1)
            out["R_vmax_unsoft"], out["Vmax_unsoft"] = r_u, v_u
            soft_r = np.maximum(softening, radius)
            r_s, v_s = get_vmax(mass, soft_r, u.G_eff)
            out["Vmax_soft"] = v_s
            # spin (subhalo_properties.py:1046-1073)
            if r_s > 0 and v_s > 0:
                mask = radius <= r_s
                vrel = velocity[mask, :] - vcom[None, :]
                Ltot = np.linalg.norm(
                    (mass[mask, None] * np.cross(position[mask, :], vrel)).sum(axis=0)
                )
                M_r = mass[mask].sum()
                if M_r > 0:
                    out["spin_parameter"] = Ltot / (np.sqrt(2.0) * M_r * v_s * r_s)
            out["EncloseRadius"] = np.max(radius)
        out["HalfMassRadiusTot"] = get_half_weight_radius(radius, mass, Mtot)
        out["HalfMassRadiusDM"] = get_half_weight_radius(radius[dm], mass[dm], M_t[1])
        baryon_kinematics(out, mass, position, velocity, types, radius, centre, u.boxsize)
        # bound particles inside 10 half-mass radii, no search-radius condition (subhalo_properties.py:1075-1125, :1302-1352)
        inertia_tensors(out, mass, position, types,
                        (10 * out["HalfMassRadiusTot"], 10 * out["HalfMassRadiusDM"],
                         10 * out["HalfMassRadiusGas"], 10 * out["HalfMassRadiusStar"]),
                        None, (Mtot != 0, M_t[1] != 0, M_t[0] != 0, M_t[4] != 0))

        for key in self.property_list:
            _store(halo_result, g, key, out.get(key))
        if self.dmo:
            _store(halo_result, g, "ExSituFraction", None)
        halo_result["_counts"] = counts


# --------------------------------------------------------------------------
# SO  (SO_properties.py)
# --------------------------------------------------------------------------
class SOProperties:
    """SO_properties.py:313-513, :532-692, :857-913, :2725-2765, :3370-3742."""

    property_list = [
        "r", "Mtot", "Ngas", "Ndm", "Nstar", "Nbh", "com", "vcom", "Vmax_soft", "R_vmax_soft",
        "Mfrac_satellites", "Mfrac_external", "Mgas", "Mdm", "Ldm", "Mstar", "Mbh_dynamical",
        "spin_parameter", "concentration_unsoft", "concentration_soft",
        "com_gas", "com_star", "vcom_gas", "vcom_star", "Lgas", "Lstar", "Lbaryons", "DtoTgas", "DtoTstar",
        *INERTIA_KEYS,
        "Nnu", "Mnu", "MnuNS", "DarkMatterMassFlowRate",
    ]  # fmt: skip

    def __init__(self, units: Units, SOval: float, type: str = "mean",
                 halo_filter="basic", filters=DEFAULT_FILTERS, dmo=True):  # fmt: skip
        if type not in ["mean", "crit", "physical", "BN98"]:
            raise AttributeError(f"Unknown SO type: {type}!")
        self.units = units
        self.type = type
        self.halo_filter = halo_filter
        self.filters = filters
        self.dmo = dmo
        # SO_properties.py:3453-3471
        self.mean_density_multiple = 1000.0
        self.critical_density_multiple = 1000.0
        self.physical_radius_mpc = 0.0
        self.virial_definition = False
        if type == "mean":
            self.mean_density_multiple = SOval
            self.virial_definition = SOval == 200
        elif type == "crit":
            self.critical_density_multiple = SOval
            self.virial_definition = SOval == 200
        elif type == "BN98":
            self.critical_density_multiple = units.virBN98
            self.virial_definition = True
        elif type == "physical":
            self.physical_radius_mpc = 0.001 * SOval
        if type in ["mean", "crit"]:
            self.name = f"SO_{SOval:.0f}_{type}"
        elif type == "physical":
            self.name = f"SO_{SOval:.0f}_kpc"
        else:
            self.name = "SO_BN98"
        # SO_properties.py:3490-3509 (physical density), converted once to comoving
        if type == "crit":
            rho = self.critical_density_multiple * units.critical_density
            self.SO_name = f"{self.critical_density_multiple:.0f}_crit"
        elif type == "mean":
            rho = self.mean_density_multiple * units.mean_density
            self.SO_name = f"{self.mean_density_multiple:.0f}_mean"
        elif type == "physical":
            rho = 0.0
            self.SO_name = f"{1000. * self.physical_radius_mpc:.0f}_kpc"
        else:
            rho = self.critical_density_multiple * units.critical_density
            self.SO_name = "BN98"
        self.reference_density = units.density_to_comoving(rho)
        self.group_name = f"SO/{self.SO_name}"

    # radius (comoving code length) used when reference_density == 0
    def physical_radius_comoving(self):
        return self.units.mpc_to_comoving(self.physical_radius_mpc)

    def calculate(self, input_halo, search_radius, data, halo_result):
        u = self.units
        g = self.group_name
        out = {}
        do_calc = get_do_calculation(self.filters, halo_result["_counts"], self.dmo)
        if input_halo["is_central"] and do_calc[self.halo_filter]:
            out = self._calculate(input_halo, search_radius, data)
        for key in self.property_list:
            _store(halo_result, g, key, out.get(key))

    def _calculate(self, input_halo, search_radius, data):
        u = self.units
        centre = input_halo["cofp"]
        index = input_halo["index"]
        # compute_basics (SO_properties.py:313-354)
        mass, radius, position, velocity, types, groupnr, fofid, softening = ([] for _ in range(8))
        for ptype in PTYPES:
            if ptype not in data:
                continue
            d = data[ptype]
            mass.append(d["Masses"])
            pos = d["Coordinates"] - centre[None, :]
            position.append(pos)
            r = np.sqrt(np.sum(pos**2, axis=1))
            radius.append(r)
            velocity.append(d["Velocities"])
            types.append(ptype * np.ones(r.shape, dtype=np.int32))
            groupnr.append(d["GroupNr_bound"])
            fofid.append(d["FOFGroupIDs"])
            softening.append(np.ones(r.shape, dtype=np.float64) * u.softening_comoving(ptype))
        mass = np.concatenate(mass)
        radius = np.concatenate(radius)
        position = np.concatenate(position)
        velocity = np.concatenate(velocity)
        types = np.concatenate(types)
        groupnr = np.concatenate(groupnr)
        fofid = np.concatenate(fofid)
        softening = np.concatenate(softening)

        # compute_SO_radius_and_mass (SO_properties.py:356-513)
        reference_density = self.reference_density
        physical_radius = self.physical_radius_comoving()
        if len(radius) == 0:
            return {}  # reference would fail on order[0]; nothing in the sphere -> zeros
        # neutrino particles (PartType6: weighted masses, may be negative) take part in the mass profile
        # and in nothing else (SO_properties.py:380-395)
        has_nu = 6 in data
        if has_nu:
            nu_mass = data[6]["Masses"] * data[6]["Weights"]
            nu_pos = data[6]["Coordinates"] - centre[None, :]
            nu_radius = np.sqrt(np.sum(nu_pos**2, axis=1))
            all_mass = np.concatenate([mass, nu_mass])
            all_r = np.concatenate([radius, nu_radius])
        else:
            all_mass, all_r = mass, radius
        order = np.argsort(all_r)
        ordered_radius = all_r[order]
        cumulative_mass = np.cumsum(all_mass[order], dtype=np.float64).astype(mass.dtype)
        cumulative_mass += u.nu_density_comoving * 4.0 / 3.0 * np.pi * ordered_radius**3
        # FOF id of the innermost non-neutrino particle (SO_properties.py:407-409)
        fof0 = fofid[order[order < radius.shape[0]][0]]
        nskip = max(1, int(np.argmax(ordered_radius > 0.0)))
        ordered_radius = ordered_radius[nskip:]
        cumulative_mass = cumulative_mass[nskip:]
        nr_parts = len(ordered_radius)
        density = cumulative_mass / (4.0 / 3.0 * np.pi * ordered_radius**3)

        SO_r = SO_mass = None
        if reference_density > 0:
            if nr_parts > 0:
                try:
                    SO_r, SO_mass, SO_volume = find_SO_radius_and_mass(
                        ordered_radius, density, cumulative_mass, reference_density,
                        mpc20=u.mpc_to_comoving(20.0),
                    )  # fmt: skip
                except SearchRadiusTooSmallError:
                    raise SearchRadiusTooSmallError("SO radius multiple was too small!")
        elif physical_radius > 0:
            SO_r = physical_radius
            SO_volume = 4.0 * np.pi / 3.0 * _qpow(SO_r, 3)
            if nr_parts > 0:
                outside_radius = ordered_radius > SO_r
                if not np.any(outside_radius):
                    SO_mass = cumulative_mass[-1]
                else:
                    i = int(np.argmax(outside_radius))
                    if i == 0:
                        SO_mass = cumulative_mass[i]
                    else:
                        r1 = ordered_radius[i - 1]
                        r2 = ordered_radius[i]
                        M1 = cumulative_mass[i - 1]
                        M2 = cumulative_mass[i]
                        SO_mass = M1 + (SO_r - r1) / (r2 - r1) * (M2 - M1)
        # documented divergence (SURVEY 8a pitfall 7): undefined SO_r/SO_mass -> no SO
        if SO_r is None or SO_mass is None:
            return {}
        SO_exists = SO_r > 0 and SO_mass > 0
        if not SO_exists:
            return {}

        is_sat = (groupnr >= 0) & (groupnr != index) & (fofid == fof0)
        is_ext = (groupnr >= 0) & (groupnr != index) & (fofid != fof0)

        sel = radius < SO_r
        N = {t: int((radius[types == t] < SO_r).sum()) for t in PTYPES}
        all_dm = types == 1   # every dark matter particle of the search sphere (flow rate through the outer shell)
        dm_all_mass, dm_all_pos, dm_all_vel = mass[all_dm], position[all_dm], velocity[all_dm]
        # members first, then the surrounding particles (SO_properties.py:491-496, :629-630)
        it_mass = np.concatenate([mass[sel], mass[~sel]])
        it_pos = np.concatenate([position[sel], position[~sel]])
        it_types = np.concatenate([types[sel], types[~sel]])
        mass = mass[sel]
        radius = radius[sel]
        position = position[sel]
        velocity = velocity[sel]
        types = types[sel]
        is_sat = is_sat[sel]
        is_ext = is_ext[sel]
        softening = softening[sel]

        out = {"r": SO_r, "Mtot": SO_mass}
        out["Ngas"], out["Ndm"], out["Nstar"], out["Nbh"] = N[0], N[1], N[4], N[5]
        nu_sum_m = nu_sum_mr = 0.0
        out["Nnu"] = 0
        if has_nu:
            # SO_properties.py:507-511, :2689-2722
            nu_sel = nu_radius < SO_r
            out["Nnu"] = int(nu_sel.sum())
            if out["Nnu"] != 0:
                out["Mnu"] = data[6]["Masses"][nu_sel].sum()
                out["MnuNS"] = (data[6]["Masses"][nu_sel] * data[6]["Weights"][nu_sel]).sum() + u.nu_density_comoving * SO_volume
                nu_sum_m = np.sum(nu_mass[nu_sel])
                nu_sum_mr = np.sum(nu_mass[nu_sel] * nu_radius[nu_sel])
        Mtotpart = mass.sum()
        frac = mass / Mtotpart if Mtotpart != 0 else None
        with np.errstate(all="ignore"):
            if Mtotpart == 0:
                frac = mass * np.nan
        out["com"] = ((frac[:, None] * position).sum(axis=0) + centre) % u.boxsize
        vcom = (frac[:, None] * velocity).sum(axis=0)
        out["vcom"] = vcom
        if Mtotpart != 0:
            soft_r = np.maximum(softening, radius)
            r_s, v_s = get_vmax(mass, soft_r, u.G_eff)
            out["R_vmax_soft"], out["Vmax_soft"] = r_s, v_s
            if v_s > 0:
                vrel = velocity - vcom[None, :]
                Ltot = np.linalg.norm((mass[:, None] * np.cross(position, vrel)).sum(axis=0))
                out["spin_parameter"] = Ltot / (np.sqrt(2.0) * Mtotpart * SO_r * v_s)
        out["Mfrac_satellites"] = mass[is_sat].sum() / SO_mass
        out["Mfrac_external"] = mass[is_ext].sum() / SO_mass
        M_t = {t: mass[types == t].sum() for t in PTYPES}
        out["Mgas"], out["Mdm"], out["Mstar"], out["Mbh_dynamical"] = M_t[0], M_t[1], M_t[4], M_t[5]
        dm = types == 1
        if M_t[1] != 0:
            fdm = mass[dm] / M_t[1]
            vcom_dm = (fdm[:, None] * velocity[dm]).sum(axis=0)
            out["Ldm"] = get_angular_momentum(mass[dm], position[dm], velocity[dm], vcom_dm)
        baryon_kinematics(out, mass, position, velocity, types, radius, centre, u.boxsize, baryon_mass_from_parts=False)
        if self.virial_definition:
            for key, rr in (("concentration_unsoft", radius),
                            ("concentration_soft", np.maximum(softening, radius))):  # fmt: skip
                if rr.shape[0] < 10:
                    continue
                # calculate_concentration (SO_properties.py:2738-2753)
                R1 = np.sum(mass * rr)
                missed_mass = SO_mass - np.sum(mass)
                if has_nu and out["Nnu"] != 0:
                    R1 += nu_sum_mr
                    missed_mass -= nu_sum_m
                R1 += np.pi * u.nu_density_comoving * _qpow(SO_r, 4)
                missed_mass -= u.nu_density_comoving * 4.0 / 3.0 * np.pi * _qpow(SO_r, 3)
                R1 += missed_mass * SO_r
                R1 /= SO_r * SO_mass
                out[key] = concentration_from_R1(R1)
        inertia_tensors(out, it_mass, it_pos, it_types, (SO_r,) * 4, search_radius,
                        (Mtotpart != 0, M_t[1] != 0, M_t[0] != 0, M_t[4] != 0), n_inner=int(sel.sum()))
        if self.virial_definition and N[1] != 0:
            # DarkMatterMassFlowRate (SO_properties.py:2919-2932): all loaded dark matter, not only r < R_SO
            out["DarkMatterMassFlowRate"] = mass_flow_rate(
                dm_all_pos, dm_all_mass, dm_all_vel, SO_r, SO_mass, search_radius,
                mass, velocity, radius, vcom, u)
        return out


def mass_flow_rate(positions, masses, velocities, SO_r, SO_mass, search_radius, so_mass, so_velocity, so_radius, vcom, u):
    """
    calculate_flow_rate(flow_type="mass", pseudo_evolve=True, hubble=False) (SO_properties.py:2816-2916):
    mass flux through shells of width 0.1 R centred on R = 0.1, 0.3, 1 R_SO, relative to the centre-of-mass
    velocity of ALL particles inside that radius (vcom_ten_percent / vcom_thirty_percent :2791-2814, vcom)
    and to the pseudo-evolution of R_SO; inflow (v_r < 0) and outflow (v_r > 0) separately.  Raises
    SearchRadiusTooSmallError when the outermost shell reaches beyond the search radius (:2849-2850).
    Lengths are comoving here, so the rates carry 1/a relative to the table unit (applied at the cast).
    """
    radii = np.sqrt(np.sum(positions**2, axis=1))
    R_fracs = [0.1, 0.3, 1]
    rates = np.zeros(6)
    for i_R, R_frac in enumerate(R_fracs):
        R = R_frac * SO_r
        dR = 0.1 * R
        if search_radius is not None and R + (dR / 2) > search_radius:
            raise SearchRadiusTooSmallError
        r_mask = (R - (dR / 2) < radii) & (radii < R + (dR / 2))
        if R_frac == 1:
            vc = vcom
        else:
            mask = so_radius < R_frac * SO_r
            if not np.sum(mask):
                vc = np.zeros(3)   # "This shouldn't happen, but HaloCentre doesn't actually need to be on top of a particle"
            else:
                vc = (so_mass[mask, None] * so_velocity[mask]).sum(axis=0) / so_mass[mask].sum()
        r_hat = positions[r_mask] / np.stack(3 * [radii[r_mask]], axis=1)
        v_r = np.sum((velocities[r_mask] - vc[None, :]) * r_hat, axis=1)
        R_dot = (2 / 3) * (u.G * SO_mass * u.H / 100) ** (1 / 3)
        R_dot *= 2 * u.Omega_g + (3 / 2) * u.Omega_m
        R_dot *= R_frac
        v_r -= R_dot
        flow_rate = masses[r_mask] * np.abs(v_r)
        rates[i_R] = np.sum(flow_rate[v_r < 0]) / dR
        rates[i_R + 3] = np.sum(flow_rate[v_r > 0]) / dR
    return rates


class RadiusMultipleSOProperties(SOProperties):
    """SO_properties.py:3797-3926"""

    def __init__(self, units, SOval, multiple, type="mean", **kw):
        if type not in ["mean", "crit"]:
            raise AttributeError("radius multiples only for type mean or crit")
        super().__init__(units, 0, "physical", **kw)
        self.SO_name = f"{multiple:.0f}xR_{SOval:.0f}_{type}"
        self.group_name = f"SO/{self.SO_name}"
        self.name = f"SO_{self.SO_name}"
        self.requested_type = type
        self.requested_SOval = SOval
        self.multiple = multiple
        self._radius_comoving = 0.0

    def physical_radius_comoving(self):
        return self._radius_comoving

    def calculate(self, input_halo, search_radius, data, halo_result):
        key = f"SO/{self.requested_SOval:.0f}_{self.requested_type}/r"
        if key not in halo_result:
            raise RuntimeError(f"Trying to obtain {key}, but it has not been calculated!")
        r32 = halo_result[key].stored[0]  # float32, comoving code length
        # :3917  physical_radius_mpc = multiple * R.to("Mpc").value: a python float times a float32 value is float32
        # (NumPy >= 2 promotion), so the radius of this SO is the FLOAT32 product -- pinned at a = 1 by
        # tests/golden/control_flow.npz (search radius handed back by process_single_halo, SO/5xR_500_crit/SORadius);
        # the comoving <-> physical factors around it are applied in float64 here
        # With a != 1 the float32 value also goes through two unit conversions, each a float32 product (unyt converts
        # a float32 array with a float32 factor): comoving -> physical Mpc before the multiplication (:3917), physical
        # Mpc -> comoving when the radius is compared with the particle radii (:425-452) or handed to the scheduler
        # (halo_tasks.py:168) -- pinned by tests/golden/control_flow_units.npz under that reading of unyt
        mpc = 1000.0 * self.units.kpc
        to_mpc32, from_mpc32 = np.float32(self.units.a / mpc), np.float32(mpc / self.units.a)
        r_mpc32 = np.float32(self.multiple) * (np.float32(r32) * to_mpc32)
        self._radius_comoving = float(r_mpc32 * from_mpc32)
        self.physical_radius_mpc = r_mpc32   # stays a float32 number: the scheduler converts it with a float32 factor
        r_mult32 = np.float32(self.multiple) * np.float32(r32)
        # :3920  `multiple * halo_result[key][0]`: a float32 product in the comoving unit of the stored radius, compared
        # with the float64 search radius
        if float(r_mult32) > float(search_radius):
            raise SearchRadiusTooSmallError("SO radius multiple estimate was too small!")
        super().calculate(input_halo, search_radius, data, halo_result)


# --------------------------------------------------------------------------
# spherical apertures (aperture_properties.py)
# --------------------------------------------------------------------------
class ApertureProperties:
    """aperture_properties.py:270-317, :373-539, :1099-1126, :1232-1281, :3469-3577,
    :3879-4244 -- DMO/basic property subset."""

    property_list = [
        "Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar", "Nbh",
        "com", "vcom", "Ldm", "veldisp_matrix_dm", "HalfMassRadiusDM", "Vmax_soft", "R_vmax_soft",
        "com_star", "vcom_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas", "veldisp_matrix_star",
        "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
        "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar",
    ]  # fmt: skip

    # output name -> key of the bound-subhalo properties an aperture radius can be a multiple of
    RADIUS_PROPERTY_KEYS = {
        "HalfMassRadiusTotal": "HalfMassRadiusTot", "HalfMassRadiusDarkMatter": "HalfMassRadiusDM",
        "HalfMassRadiusGas": "HalfMassRadiusGas", "HalfMassRadiusStars": "HalfMassRadiusStar",
        "HalfMassRadiusBaryons": "HalfMassRadiusBaryon", "EncloseRadius": "EncloseRadius",
    }  # fmt: skip

    def __init__(self, units: Units, radius_kpc: Optional[float], inclusive: bool, all_radii_kpc=None,
                 halo_filter="basic", filters=DEFAULT_FILTERS, dmo=True, aperture_property=None):  # fmt: skip
        """aperture_property = ("BoundSubhalo/<output name>", multiple) instead of radius_kpc: aperture_properties.py:3973-3981"""
        self.units = units
        self.aperture_physical_radius_kpc = radius_kpc
        self.aperture_property = aperture_property
        self.inclusive = inclusive
        self.all_radii_kpc = list(all_radii_kpc) if all_radii_kpc is not None else [radius_kpc]
        self.halo_filter = halo_filter
        self.filters = filters
        self.dmo = dmo
        self.mean_density_multiple = None
        self.critical_density_multiple = None
        if radius_kpc is not None:
            self.physical_radius_mpc = 0.001 * radius_kpc
            aperture_name = f"{1000*radius_kpc:.0f}pc" if radius_kpc < 1 else f"{radius_kpc:.0f}kpc"
        else:
            prop = aperture_property[0].split("/")[-1]
            assert aperture_property[0].split("/")[0] == "BoundSubhalo"
            aperture_name = prop if aperture_property[1] == 1 else f"{int(aperture_property[1])}x{prop}"
            self.physical_radius_mpc = 0
            self._radius_key = f"BoundSubhalo/{self.RADIUS_PROPERTY_KEYS[prop]}"
        if inclusive:
            self.name = f"inclusive_sphere_{aperture_name}"
            self.group_name = f"InclusiveSphere/{aperture_name}"
        else:
            self.name = f"exclusive_sphere_{aperture_name}"
            self.group_name = f"ExclusiveSphere/{aperture_name}"

    def calculate(self, input_halo, search_radius, data, halo_result):
        u = self.units
        g = self.group_name
        do_calc = get_do_calculation(self.filters, halo_result["_counts"], self.dmo)
        out = {}
        copied = None
        skip = False
        # skip/copy rule (aperture_properties.py:4082-4123)
        i_radius = self.all_radii_kpc.index(self.aperture_physical_radius_kpc) if self.aperture_property is None else 0
        if i_radius != 0 and ("BoundSubhalo/EncloseRadius" in halo_result):
            r_enclose = float(halo_result["BoundSubhalo/EncloseRadius"].stored[0])  # float32
            r_previous_kpc = self.all_radii_kpc[i_radius - 1]
            if u.kpc_to_comoving(r_previous_kpc) > r_enclose:
                skip = True
                if (not self.inclusive) and do_calc[self.halo_filter]:
                    prev = (f"{1000*r_previous_kpc:.0f}pc" if r_previous_kpc < 1
                            else f"{r_previous_kpc:.0f}kpc")  # fmt: skip
                    copied = f"ExclusiveSphere/{prev}"
        if do_calc[self.halo_filter] and not skip:
            if self.aperture_property is None:
                aperture_radius = u.kpc_to_comoving(self.aperture_physical_radius_kpc)
            else:
                # :4131-4138  multiple * halo_result[property][0]: a python number times the stored float32 value is a
                # float32 product (NumPy >= 2); both sides of the comparisons below are comoving lengths
                if self._radius_key not in halo_result:
                    raise RuntimeError(f"{self.aperture_property[0]} must be enabled if you want to use it to define an aperture")
                aperture_radius = float(np.float32(self.aperture_property[1]) * halo_result[self._radius_key].stored[0])
            if search_radius < aperture_radius:
                raise SearchRadiusTooSmallError("Search radius is smaller than aperture")
            out = self._calculate(input_halo, aperture_radius, data)
        if self.dmo:
            _store(halo_result, g, "ExSituFraction", None)
        for key in self.property_list:
            if copied is not None:
                pv = halo_result[f"{copied}/{key}"]
                halo_result[f"{g}/{key}"] = Value(pv.raw.copy(), pv.stored.copy())
            else:
                _store(halo_result, g, key, out.get(key))

    def _calculate(self, input_halo, aperture_radius, data):
        u = self.units
        centre = input_halo["cofp"]
        index = input_halo["index"]
        mass, position, radius, velocity, types, softening = ([] for _ in range(6))
        for ptype in PTYPES:
            if ptype not in data:
                continue
            d = data[ptype]
            grnr = d["GroupNr_bound"]
            in_halo = np.ones(grnr.shape, dtype=bool) if self.inclusive else grnr == index
            mass.append(d["Masses"][in_halo])
            pos = d["Coordinates"][in_halo, :] - centre[None, :]
            position.append(pos)
            r = np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2 + pos[:, 2] ** 2)
            radius.append(r)
            velocity.append(d["Velocities"][in_halo, :])
            types.append(ptype * np.ones(r.shape, dtype=np.int32))
            softening.append(np.ones(r.shape, dtype=np.float64) * u.softening_comoving(ptype))
        mass = np.concatenate(mass)
        position = np.concatenate(position)
        radius = np.concatenate(radius)
        velocity = np.concatenate(velocity)
        types_all = np.concatenate(types)
        softening = np.concatenate(softening)
        mask = radius <= aperture_radius
        mass = mass[mask]
        position = position[mask]
        velocity = velocity[mask]
        radius = radius[mask]
        type_ = types_all[mask]
        softening = softening[mask]

        out = {}
        N = {t: int(mask[types_all == t].sum()) for t in PTYPES}
        out["Ngas"], out["Ndm"], out["Nstar"], out["Nbh"] = N[0], N[1], N[4], N[5]
        Mtot = mass.sum()
        out["Mtot"] = Mtot
        M_t = {t: mass[type_ == t].sum() for t in PTYPES}
        out["Mgas"], out["Mdm"], out["Mstar"], out["Mbh_dynamical"] = M_t[0], M_t[1], M_t[4], M_t[5]
        if Mtot != 0:
            frac = mass / Mtot
            out["com"] = ((frac[:, None] * position).sum(axis=0) + centre) % u.boxsize
            out["vcom"] = (frac[:, None] * velocity).sum(axis=0)
            soft_r = np.maximum(softening, radius)
            out["R_vmax_soft"], out["Vmax_soft"] = get_vmax(mass, soft_r, u.G_eff)
        dm = type_ == 1
        if M_t[1] != 0:
            fdm = mass[dm] / M_t[1]
            vcom_dm = (fdm[:, None] * velocity[dm]).sum(axis=0)
            out["veldisp_matrix_dm"] = get_velocity_dispersion_matrix(fdm, velocity[dm], vcom_dm)[0]
            out["Ldm"] = get_angular_momentum(mass[dm], position[dm], velocity[dm], vcom_dm)
        out["HalfMassRadiusDM"] = get_half_weight_radius(radius[dm], mass[dm], M_t[1])
        baryon_kinematics(out, mass, position, velocity, type_, radius, centre, u.boxsize)
        return out


# --------------------------------------------------------------------------
# projected apertures (projected_aperture_properties.py)
# --------------------------------------------------------------------------
class ProjectedApertureProperties:
    """projected_aperture_properties.py:98-142, :172-198, :254-378, :739-786, :946-957,
    :1491-1577, :1798-2002 -- DMO/basic subset; bound particles only."""

    property_list = ["Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar",
                     "Nbh", "com", "vcom", "HalfMassRadiusDM", "proj_veldisp_dm",
                     "com_star", "proj_veldisp_gas", "proj_veldisp_star",
                     "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
                     *PROJECTED_INERTIA_KEYS]  # fmt: skip

    # properties with a direct dependence on the aperture size (property_names, :1596-1676, value True) among the ones
    # restated here: a copied aperture leaves them at zero when strict_halo_copy is set (:1880-1883)
    APERTURE_DEPENDENT = tuple(k for k in PROJECTED_INERTIA_KEYS if "Noniterative" not in k)

    def __init__(self, units: Units, radius_kpc: float, all_radii_kpc=None,
                 halo_filter="basic", filters=DEFAULT_FILTERS, dmo=True, strict_halo_copy=False):  # fmt: skip
        self.units = units
        self.aperture_physical_radius_kpc = radius_kpc
        self.all_radii_kpc = list(all_radii_kpc) if all_radii_kpc is not None else [radius_kpc]
        self.halo_filter = halo_filter
        self.filters = filters
        self.dmo = dmo
        self.strict_halo_copy = strict_halo_copy
        self.physical_radius_mpc = 0.001 * radius_kpc
        self.mean_density_multiple = None
        self.critical_density_multiple = None
        aperture_name = f"{1000*radius_kpc:.0f}pc" if radius_kpc < 1 else f"{radius_kpc:.0f}kpc"
        self.name = f"projected_aperture_{aperture_name}"
        self.group_name = f"ProjectedAperture/{aperture_name}"

    def calculate(self, input_halo, search_radius, data, halo_result):
        u = self.units
        do_calc = get_do_calculation(self.filters, halo_result["_counts"], self.dmo)
        skip = False
        copied = None
        i_radius = self.all_radii_kpc.index(self.aperture_physical_radius_kpc)
        if i_radius != 0 and ("BoundSubhalo/EncloseRadius" in halo_result):
            r_enclose = float(halo_result["BoundSubhalo/EncloseRadius"].stored[0])
            r_previous_kpc = self.all_radii_kpc[i_radius - 1]
            if u.kpc_to_comoving(r_previous_kpc) > r_enclose:
                skip = True
                if do_calc[self.halo_filter]:
                    prev = (f"{1000*r_previous_kpc:.0f}pc" if r_previous_kpc < 1
                            else f"{r_previous_kpc:.0f}kpc")  # fmt: skip
                    copied = f"ProjectedAperture/{prev}"
        outs = {p: {} for p in ("projx", "projy", "projz")}
        if do_calc[self.halo_filter] and not skip:
            aperture_radius = u.kpc_to_comoving(self.aperture_physical_radius_kpc)
            outs = self._calculate(input_halo, aperture_radius, data)
        for proj in ("projx", "projy", "projz"):
            g = f"{self.group_name}/{proj}"
            for key in self.property_list:
                if copied is not None and not (self.strict_halo_copy and key in self.APERTURE_DEPENDENT):
                    pv = halo_result[f"{copied}/{proj}/{key}"]
                    halo_result[f"{g}/{key}"] = Value(pv.raw.copy(), pv.stored.copy())
                else:
                    _store(halo_result, g, key, outs[proj].get(key))

    def _calculate(self, input_halo, aperture_radius, data):
        u = self.units
        centre = input_halo["cofp"]
        index = input_halo["index"]
        mass, position, velocity, types = ([] for _ in range(4))
        rp = ([], [], [])
        for ptype in PTYPES:
            if ptype not in data:
                continue
            d = data[ptype]
            in_halo = d["GroupNr_bound"] == index
            mass.append(d["Masses"][in_halo])
            pos = d["Coordinates"][in_halo, :] - centre[None, :]
            position.append(pos)
            rp[0].append(np.sqrt(pos[:, 1] ** 2 + pos[:, 2] ** 2))
            rp[1].append(np.sqrt(pos[:, 0] ** 2 + pos[:, 2] ** 2))
            rp[2].append(np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2))
            velocity.append(d["Velocities"][in_halo, :])
            types.append(ptype * np.ones(pos.shape[0], dtype=np.int32))
        mass = np.concatenate(mass)
        position = np.concatenate(position)
        velocity = np.concatenate(velocity)
        types = np.concatenate(types)
        outs = {}
        for iproj, proj in enumerate(("projx", "projy", "projz")):
            projr = np.concatenate(rp[iproj])
            pm = projr <= aperture_radius
            m = mass[pm]
            x = position[pm]
            v = velocity[pm]
            r = projr[pm]
            t = types[pm]
            out = {}
            N = {tt: int((t == tt).sum()) for tt in PTYPES}
            out["Ngas"], out["Ndm"], out["Nstar"], out["Nbh"] = N[0], N[1], N[4], N[5]
            Mtot = m.sum()
            out["Mtot"] = Mtot
            M_t = {tt: m[t == tt].sum() for tt in PTYPES}
            out["Mgas"], out["Mdm"], out["Mstar"], out["Mbh_dynamical"] = (
                M_t[0], M_t[1], M_t[4], M_t[5])  # fmt: skip
            if Mtot != 0:
                frac = m / Mtot
                out["com"] = ((frac[:, None] * x).sum(axis=0) + centre) % u.boxsize
                out["vcom"] = (frac[:, None] * v).sum(axis=0)
            dm = t == 1
            if M_t[1] != 0:
                fdm = m[dm] / M_t[1]
                pv = v[dm, iproj]
                vc = (fdm * pv).sum()
                out["proj_veldisp_dm"] = np.sqrt((fdm * (pv - vc) ** 2).sum())
            out["HalfMassRadiusDM"] = get_half_weight_radius(r[dm], m[dm], M_t[1])
            # gas / stars / baryons in projection (projected_aperture_properties.py:739-786, :864-986,
            # :1491-1577): star centre of mass, line-of-sight dispersions, half-mass radii in 2D radius
            gas, star = t == 0, t == 4
            if M_t[4] != 0:
                fs = m[star] / M_t[4]
                out["com_star"] = ((fs[:, None] * x[star]).sum(axis=0) + centre) % u.boxsize
            for key, sel2, Mk in (("gas", gas, M_t[0]), ("star", star, M_t[4])):
                if Mk != 0:
                    fk = m[sel2] / Mk
                    pvk = v[sel2, iproj]
                    vck = (fk * pvk).sum()
                    out[f"proj_veldisp_{key}"] = np.sqrt((fk * (pvk - vck) ** 2).sum())
            bar = gas | star
            out["HalfMassRadiusGas"] = get_half_weight_radius(r[gas], m[gas], M_t[0])
            out["HalfMassRadiusStar"] = get_half_weight_radius(r[star], m[star], M_t[4])
            out["HalfMassRadiusBaryon"] = get_half_weight_radius(r[bar], m[bar], m[bar].sum())
            # projected inertia tensors (projected_aperture_properties.py:788-850, :876-940, :1005-1068): iterative
            # over ALL bound particles of the component, fixed circle over those inside the aperture
            if ENABLE_INERTIA:
                for comp, all_sel, in_sel, Mk in (("Total", np.ones(len(mass), dtype=bool), np.ones(len(m), dtype=bool), Mtot),
                                                  ("Gas", types == 0, gas, M_t[0]), ("Stellar", types == 4, star, M_t[4])):
                    if Mk == 0:
                        continue
                    for suffix, kw in (("", {}), ("Reduced", {"reduced": True})):
                        out[f"Projected{comp}InertiaTensor{suffix}"] = get_weighted_projected_inertia_tensor(
                            mass[all_sel], position[all_sel], iproj, aperture_radius, **kw)
                        out[f"Projected{comp}InertiaTensor{suffix}Noniterative"] = get_weighted_projected_inertia_tensor(
                            m[in_sel], x[in_sel], iproj, aperture_radius, max_iterations=1, **kw)
            outs[proj] = out
        return outs


# --------------------------------------------------------------------------
# L2 scheduler (SOAP/core/halo_tasks.py)
# --------------------------------------------------------------------------
def target_density_of(halo_prop_list, units: Units) -> Optional[float]:
    """halo_tasks.py:306-317; returned in comoving code density."""
    target = None
    for hp in halo_prop_list:
        if hp.mean_density_multiple is not None:
            d = hp.mean_density_multiple * units.mean_density
            if target is None or d < target:
                target = d
        if hp.critical_density_multiple is not None:
            d = hp.critical_density_multiple * units.critical_density
            if target is None or d < target:
                target = d
    return None if target is None else units.density_to_comoving(target)


def process_single_halo(mesh, data, halo_prop_list, units: Units, input_halo, target_density,
                        stats=None):  # fmt: skip
    """
    halo_tasks.py:23-187.  data[ptype] = dict(Coordinates, Velocities, Masses,
    GroupNr_bound, FOFGroupIDs).  Returns halo_result (dict) or None when a larger
    region must be read (input_halo["search_radius"] is then updated, as in the
    reference).  `stats` (optional dict) receives n_loop, n_sel and n_cand of the
    final iteration (for the particle-aperture-evaluation metric).
    """
    boxsize = units.boxsize
    halo_prop_done = np.zeros(len(halo_prop_list), dtype=bool)
    halo_result = {}
    current_radius = input_halo["search_radius"]
    n_loop = 0
    while True:
        n_loop += 1
        assert current_radius <= input_halo["read_radius"]
        mass_total = 0.0
        idx = {}
        ncand = 0
        for ptype in data:
            pos = data[ptype]["Coordinates"]
            idx[ptype], nc = mesh[ptype].query_radius_periodic(
                input_halo["cofp"], current_radius, pos, boxsize, return_ncand=True
            )
            ncand += nc
            if ptype in PTYPES_FOR_SO_MASSES:
                mass_total += np.sum(data[ptype]["Masses"][idx[ptype]], dtype=float)
        density = mass_total / (4.0 / 3.0 * np.pi * _qpow(current_radius, 3))
        max_physical_radius_mpc = 0.0
        if target_density is None or density <= target_density:
            particle_data = {}
            for ptype in data:
                particle_data[ptype] = {}
                for name in data[ptype]:
                    particle_data[ptype][name] = data[ptype][name][idx[ptype], ...]
            for ptype in particle_data:
                pos = particle_data[ptype]["Coordinates"]
                offset = input_halo["cofp"] - 0.5 * boxsize
                pos[:, :] = ((pos - offset) % boxsize) + offset
            for prop_nr, halo_prop in enumerate(halo_prop_list):
                if halo_prop_done[prop_nr]:
                    continue
                try:
                    halo_prop.calculate(input_halo, current_radius, particle_data, halo_result)
                except SearchRadiusTooSmallError:
                    max_physical_radius_mpc = max(
                        max_physical_radius_mpc, halo_prop.physical_radius_mpc
                    )
                    break
                else:
                    halo_prop_done[prop_nr] = True
            if np.all(halo_prop_done):
                if stats is not None:
                    stats["n_loop"] = n_loop
                    stats["n_sel"] = int(sum(len(v) for v in idx.values()))
                    stats["n_cand"] = int(ncand)
                    stats["final_radius"] = float(current_radius)
                break
        search_radius = input_halo["search_radius"]
        if isinstance(max_physical_radius_mpc, np.float32):
            # a radius multiple asked for it (halo_tasks.py:168 converts the float32 quantity with a float32 factor)
            required_radius = float(max_physical_radius_mpc * np.float32(1000.0 * units.kpc / units.a))
        else:
            required_radius = units.mpc_to_comoving(max_physical_radius_mpc)
        if stats is not None:
            stats["n_loop"] = n_loop
        if required_radius > input_halo["read_radius"]:
            input_halo["search_radius"] = max(search_radius, required_radius)
            return None
        elif current_radius >= input_halo["read_radius"]:
            input_halo["search_radius"] = max(search_radius, current_radius)
            return None
        else:
            current_radius = min(current_radius * SEARCH_RADIUS_FACTOR, input_halo["read_radius"])
            current_radius = max(current_radius, required_radius)
    return halo_result


def process_halos(mesh, data, halo_prop_list, units: Units, halo_arrays, stats_list=None):
    """
    halo_tasks.py:276-430, single rank.  halo_arrays: dict of arrays (index, cofp,
    search_radius, read_radius, is_central, nr_bound_part, done).  Halos are visited
    in array order (the caller sorts by nr_bound_part descending, chunk_tasks.py:118).
    Returns the list of (halo position in arrays, halo_result).
    """
    target_density = target_density_of(halo_prop_list, units)
    results = []
    nr_halos = len(halo_arrays["index"])
    for i in range(nr_halos):
        if halo_arrays["done"][i] != 0:
            continue
        input_halo = {name: halo_arrays[name][i, ...].copy() for name in halo_arrays}
        stats = {} if stats_list is not None else None
        halo_result = process_single_halo(
            mesh, data, halo_prop_list, units, input_halo,
            target_density if input_halo["is_central"] == 1 else None, stats,
        )  # fmt: skip
        if halo_result is not None:
            halo_result.pop("_counts", None)
            results.append((i, halo_result))
            halo_arrays["done"][i] = 1
            if stats_list is not None:
                stats_list.append(stats)
        else:
            new_read_radius = max(
                input_halo["read_radius"] * READ_RADIUS_FACTOR, input_halo["search_radius"]
            )
            halo_arrays["read_radius"][i] = new_read_radius
            halo_arrays["search_radius"][i] = input_halo["search_radius"]
    return results
