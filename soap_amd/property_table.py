"""
Output schema of the in-scope halo properties: our own restatement of the rows of
SOAP/property_table.py `PropertyTable.full_property_list` that the hot path produces
(SURVEY.md section 8(a) row a22).  Key = the reference's internal property key.

Fields: output name, number of components, stored dtype, unit string, whether the
stored value is physical (False = comoving, multiply the unit by a**a_exponent),
a-scale exponent for comoving quantities, and whether a DMO run emits it.
`a_power` is the power of the expansion factor that converts the engine's raw value
(comoving lengths, code velocities/masses) into the stored unit.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass(frozen=True)
class Property:
    name: str
    shape: int
    dtype: type
    unit: str
    output_physical: bool
    a_scale_exponent: int | None
    dmo_property: bool
    a_power: int = 0  # stored = raw * a**a_power
    description: str = ""


_M = "snap_mass"
_Lc = "snap_length"
_V = "snap_length/snap_time"
_L = "snap_mass*snap_length**2/snap_time"
_V2 = "snap_length**2/snap_time**2"

full_property_list = {
    # key: Property(name, shape, dtype, unit, physical, a_exp, dmo, a_power, description)
    "r": Property("SORadius", 1, np.float32, _Lc, False, 1, True, 0, "Radius of the sphere {label}."),
    "Mtot": Property("TotalMass", 1, np.float32, _M, True, 0, True, 0, "Total mass."),
    "Mgas": Property("GasMass", 1, np.float32, _M, True, 0, False, 0, "Mass in gas particles."),
    "Mdm": Property("DarkMatterMass", 1, np.float32, _M, True, 0, True, 0, "Mass in dark matter particles."),
    "Mstar": Property("StellarMass", 1, np.float32, _M, True, 0, False, 0, "Mass in star particles."),
    "Mbh_dynamical": Property("BlackHolesDynamicalMass", 1, np.float32, _M, True, 0, False, 0, "Dynamical mass in black holes."),
    "Ngas": Property("NumberOfGasParticles", 1, np.uint32, "dimensionless", True, None, False, 0, "Number of gas particles."),
    "Ndm": Property("NumberOfDarkMatterParticles", 1, np.uint32, "dimensionless", True, None, True, 0, "Number of dark matter particles."),
    "Nstar": Property("NumberOfStarParticles", 1, np.uint32, "dimensionless", True, None, False, 0, "Number of star particles."),
    "Nbh": Property("NumberOfBlackHoleParticles", 1, np.uint32, "dimensionless", True, None, False, 0, "Number of black hole particles."),
    "com": Property("CentreOfMass", 3, np.float64, _Lc, False, 1, True, 0, "Centre of mass."),
    "vcom": Property("CentreOfMassVelocity", 3, np.float32, _V, True, 0, True, 0, "Centre of mass velocity."),
    "Ldm": Property("AngularMomentumDarkMatter", 3, np.float32, _L, True, 1, True, 1, "Angular momentum of the dark matter about the dark matter centre of mass velocity."),
    "veldisp_matrix_dm": Property("DarkMatterVelocityDispersionMatrix", 6, np.float32, _V2, True, 0, True, 0, "Mass-weighted velocity dispersion of the dark matter (XX YY ZZ XY XZ YZ)."),
    "proj_veldisp_dm": Property("DarkMatterProjectedVelocityDispersion", 1, np.float32, _V, True, 0, True, 0, "Dark matter velocity dispersion along the projection axis."),
    "HalfMassRadiusDM": Property("HalfMassRadiusDarkMatter", 1, np.float32, _Lc, False, 1, True, 0, "Dark matter half mass radius."),
    "HalfMassRadiusTot": Property("HalfMassRadiusTotal", 1, np.float32, _Lc, False, 1, True, 0, "Total half mass radius."),
    "EncloseRadius": Property("EncloseRadius", 1, np.float32, _Lc, False, 1, True, 0, "Radius of the outermost bound particle."),
    "Vmax_soft": Property("MaximumCircularVelocity", 1, np.float32, _V, True, 0, True, 0, "Maximum circular velocity, radii floored at the softening length."),
    "R_vmax_soft": Property("MaximumCircularVelocityRadius", 1, np.float32, _Lc, False, 1, True, 0, "Radius of the maximum circular velocity (softened radii)."),
    "Vmax_unsoft": Property("MaximumCircularVelocityUnsoftened", 1, np.float32, _V, True, 0, True, 0, "Maximum circular velocity, raw radii."),
    "R_vmax_unsoft": Property("MaximumCircularVelocityRadiusUnsoftened", 1, np.float32, _Lc, False, 1, True, 0, "Radius of the maximum circular velocity (raw radii)."),
    "spin_parameter": Property("SpinParameter", 1, np.float32, "dimensionless", True, 0, True, 0, "Bullock et al. (2001) spin parameter."),
    "concentration_soft": Property("Concentration", 1, np.float32, "dimensionless", True, 0, True, 0, "NFW concentration from the first radial moment (softened radii)."),
    "concentration_unsoft": Property("ConcentrationUnsoftened", 1, np.float32, "dimensionless", True, 0, True, 0, "NFW concentration from the first radial moment (raw radii)."),
    "Mfrac_satellites": Property("MassFractionSatellites", 1, np.float32, "dimensionless", True, 0, True, 0, "Mass fraction bound to satellites in the same FOF group."),
    "Mfrac_external": Property("MassFractionExternal", 1, np.float32, "dimensionless", True, 0, True, 0, "Mass fraction bound to subhalos outside this FOF group."),
    # gas / star / baryon kinematics (SOAP/property_table.py rows :1329-1448, :1541-1631, :3646-3669, :3856-3868, :4025-4073)
    "com_gas": Property("GasCentreOfMass", 3, np.float64, _Lc, False, 1, False, 0, "Centre of mass of the gas."),
    "com_star": Property("StellarCentreOfMass", 3, np.float64, _Lc, False, 1, False, 0, "Centre of mass of the stars."),
    "vcom_gas": Property("GasCentreOfMassVelocity", 3, np.float32, _V, True, 0, False, 0, "Centre of mass velocity of the gas."),
    "vcom_star": Property("StellarCentreOfMassVelocity", 3, np.float32, _V, True, 0, False, 0, "Centre of mass velocity of the stars."),
    "Lgas": Property("AngularMomentumGas", 3, np.float32, _L, True, 1, False, 1, "Angular momentum of the gas about the gas centre of mass velocity."),
    "Lstar": Property("AngularMomentumStars", 3, np.float32, _L, True, 1, False, 1, "Angular momentum of the stars about the stellar centre of mass velocity."),
    "Lbaryons": Property("AngularMomentumBaryons", 3, np.float32, _L, True, 1, False, 1, "Angular momentum of gas and stars about the baryonic centre of mass velocity."),
    "veldisp_matrix_gas": Property("GasVelocityDispersionMatrix", 6, np.float32, _V2, True, 0, False, 0, "Mass-weighted velocity dispersion of the gas (XX YY ZZ XY XZ YZ)."),
    "veldisp_matrix_star": Property("StellarVelocityDispersionMatrix", 6, np.float32, _V2, True, 0, False, 0, "Mass-weighted velocity dispersion of the stars (XX YY ZZ XY XZ YZ)."),
    "proj_veldisp_gas": Property("GasProjectedVelocityDispersion", 1, np.float32, _V, True, 0, False, 0, "Gas velocity dispersion along the projection axis."),
    "proj_veldisp_star": Property("StellarProjectedVelocityDispersion", 1, np.float32, _V, True, 0, False, 0, "Stellar velocity dispersion along the projection axis."),
    "HalfMassRadiusGas": Property("HalfMassRadiusGas", 1, np.float32, _Lc, False, 1, False, 0, "Gas half mass radius."),
    "HalfMassRadiusStar": Property("HalfMassRadiusStars", 1, np.float32, _Lc, False, 1, False, 0, "Stellar half mass radius."),
    "Nnu": Property("NumberOfNeutrinoParticles", 1, np.uint32, "dimensionless", True, None, False, 0, "Number of neutrino particles."),
    "Mnu": Property("RawNeutrinoMass", 1, np.float32, _M, True, 0, True, 0, "Total mass of the neutrino particles, without their weights."),
    "MnuNS": Property("NoiseSuppressedNeutrinoMass", 1, np.float32, _M, True, 0, True, 0, "Weighted neutrino particle mass plus the mean neutrino density times the SO volume."),
    "TotalInertiaTensor": Property("TotalInertiaTensor", 6, np.float32, "snap_length**2", True, 2, True, 2, "3D inertia tensor of the total mass distribution relative to the halo centre (computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "TotalInertiaTensorReduced": Property("TotalInertiaTensorReduced", 6, np.float32, "dimensionless", True, 0, True, 0, "3D inertia tensor of the total mass distribution relative to the halo centre (reduced, computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "TotalInertiaTensorNoniterative": Property("TotalInertiaTensorNoniterative", 6, np.float32, "snap_length**2", True, 2, True, 2, "3D inertia tensor of the total mass distribution relative to the halo centre (from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "TotalInertiaTensorReducedNoniterative": Property("TotalInertiaTensorReducedNoniterative", 6, np.float32, "dimensionless", True, 0, True, 0, "3D inertia tensor of the total mass distribution relative to the halo centre (reduced, from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "DarkMatterInertiaTensor": Property("DarkMatterInertiaTensor", 6, np.float32, "snap_length**2", True, 2, True, 2, "3D inertia tensor of the dark matter distribution relative to the halo centre (computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "DarkMatterInertiaTensorReduced": Property("DarkMatterInertiaTensorReduced", 6, np.float32, "dimensionless", True, 0, True, 0, "3D inertia tensor of the dark matter distribution relative to the halo centre (reduced, computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "DarkMatterInertiaTensorNoniterative": Property("DarkMatterInertiaTensorNoniterative", 6, np.float32, "snap_length**2", True, 2, True, 2, "3D inertia tensor of the dark matter distribution relative to the halo centre (from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "DarkMatterInertiaTensorReducedNoniterative": Property("DarkMatterInertiaTensorReducedNoniterative", 6, np.float32, "dimensionless", True, 0, True, 0, "3D inertia tensor of the dark matter distribution relative to the halo centre (reduced, from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "GasInertiaTensor": Property("GasInertiaTensor", 6, np.float32, "snap_length**2", True, 2, False, 2, "3D inertia tensor of the gas mass distribution relative to the halo centre (computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "GasInertiaTensorReduced": Property("GasInertiaTensorReduced", 6, np.float32, "dimensionless", True, 0, False, 0, "3D inertia tensor of the gas mass distribution relative to the halo centre (reduced, computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "GasInertiaTensorNoniterative": Property("GasInertiaTensorNoniterative", 6, np.float32, "snap_length**2", True, 2, False, 2, "3D inertia tensor of the gas mass distribution relative to the halo centre (from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "GasInertiaTensorReducedNoniterative": Property("GasInertiaTensorReducedNoniterative", 6, np.float32, "dimensionless", True, 0, False, 0, "3D inertia tensor of the gas mass distribution relative to the halo centre (reduced, from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "StellarInertiaTensor": Property("StellarInertiaTensor", 6, np.float32, "snap_length**2", True, 2, False, 2, "3D inertia tensor of the stellar mass distribution relative to the halo centre (computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "StellarInertiaTensorReduced": Property("StellarInertiaTensorReduced", 6, np.float32, "dimensionless", True, 0, False, 0, "3D inertia tensor of the stellar mass distribution relative to the halo centre (reduced, computed iteratively): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "StellarInertiaTensorNoniterative": Property("StellarInertiaTensorNoniterative", 6, np.float32, "snap_length**2", True, 2, False, 2, "3D inertia tensor of the stellar mass distribution relative to the halo centre (from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "StellarInertiaTensorReducedNoniterative": Property("StellarInertiaTensorReducedNoniterative", 6, np.float32, "dimensionless", True, 0, False, 0, "3D inertia tensor of the stellar mass distribution relative to the halo centre (reduced, from the particles inside the initial sphere): (1,1), (2,2), (3,3), (1,2), (1,3), (2,3). Only calculated with at least 20 particles."),
    "ProjectedTotalInertiaTensor": Property("ProjectedTotalInertiaTensor", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the total mass distribution in projection (computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedTotalInertiaTensorReduced": Property("ProjectedTotalInertiaTensorReduced", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the total mass distribution in projection (reduced, computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedTotalInertiaTensorNoniterative": Property("ProjectedTotalInertiaTensorNoniterative", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the total mass distribution in projection (from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ProjectedTotalInertiaTensorReducedNoniterative": Property("ProjectedTotalInertiaTensorReducedNoniterative", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the total mass distribution in projection (reduced, from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ProjectedGasInertiaTensor": Property("ProjectedGasInertiaTensor", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the gas mass distribution in projection (computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedGasInertiaTensorReduced": Property("ProjectedGasInertiaTensorReduced", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the gas mass distribution in projection (reduced, computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedGasInertiaTensorNoniterative": Property("ProjectedGasInertiaTensorNoniterative", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the gas mass distribution in projection (from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ProjectedGasInertiaTensorReducedNoniterative": Property("ProjectedGasInertiaTensorReducedNoniterative", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the gas mass distribution in projection (reduced, from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ProjectedStellarInertiaTensor": Property("ProjectedStellarInertiaTensor", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the stellar mass distribution in projection (computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedStellarInertiaTensorReduced": Property("ProjectedStellarInertiaTensorReduced", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the stellar mass distribution in projection (reduced, computed iteratively): (1,1), (2,2), (1,2)."),
    "ProjectedStellarInertiaTensorNoniterative": Property("ProjectedStellarInertiaTensorNoniterative", 3, np.float32, "snap_length**2", True, 2, False, 2, "2D inertia tensor of the stellar mass distribution in projection (from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ProjectedStellarInertiaTensorReducedNoniterative": Property("ProjectedStellarInertiaTensorReducedNoniterative", 3, np.float32, "dimensionless", True, 0, False, 0, "2D inertia tensor of the stellar mass distribution in projection (reduced, from the bound particles inside the projected aperture): (1,1), (2,2), (1,2)."),
    "ExSituFraction": Property("ExSituFraction", 1, np.float32, "dimensionless", True, 0, True, 0, "Mass fraction of bound stars that formed in a different subhalo (zero without stars)."),
    "kappa_corot_gas": Property("KappaCorotGas", 1, np.float32, "dimensionless", True, 0, False, 0, "Kinetic energy fraction of the gas in ordered co-rotation (Correa et al. 2017)."),
    "kappa_corot_star": Property("KappaCorotStars", 1, np.float32, "dimensionless", True, 0, False, 0, "Kinetic energy fraction of the stars in ordered co-rotation."),
    "kappa_corot_baryons": Property("KappaCorotBaryons", 1, np.float32, "dimensionless", True, 0, False, 0, "Kinetic energy fraction of gas and stars in ordered co-rotation."),
    "DtoTgas": Property("DiscToTotalGasMassFraction", 1, np.float32, "dimensionless", True, 0, False, 0, "Fraction of the gas mass in the disc: 1 - 2 * counter-rotating mass fraction."),
    "DtoTstar": Property("DiscToTotalStellarMassFraction", 1, np.float32, "dimensionless", True, 0, False, 0, "Fraction of the stellar mass in the disc: 1 - 2 * counter-rotating mass fraction."),
    "DarkMatterMassFlowRate": Property("DarkMatterMassFlowRate", 6, np.float32, "snap_mass/snap_time", True, 0, True, -1, "Mass flow rate of dark matter particles through spherical shells. Contains 6 entries: inflow rate at 0.1R, 0.3R, R, outflow rate at 0.1R, 0.3R, R."),
    "HalfMassRadiusBaryon": Property("HalfMassRadiusBaryons", 1, np.float32, _Lc, False, 1, False, 0, "Baryonic (gas and stars) half mass radius."),
}  # fmt: skip


def stored_value(key, raw, a):
    """The value the reference would store: zero-initialised table-dtype array += value in
    the table unit (SO_properties.py:3600-3620, :3682-3699), i.e. one cast."""
    prop = full_property_list[key]
    raw = np.atleast_1d(np.asarray(raw, dtype=np.float64))
    if np.issubdtype(prop.dtype, np.integer):
        return raw.astype(prop.dtype)
    out = np.zeros(raw.shape, dtype=prop.dtype)
    out += raw * (a**prop.a_power) if prop.a_power else raw
    return out
