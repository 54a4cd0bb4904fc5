"""
Column layout of the engine's fp64 result table; mirrors soap_amd/csrc/layout.h
(tests/test_abi.py checks both against soap_engine_calc_columns()).

Keys are the reference's internal property keys (SOAP/property_table.py
`full_property_list` keys); values are (first column, number of columns).
"""

CALC_SUBHALO = 0
CALC_SO_DENSITY = 1
CALC_SO_PHYSICAL = 2
CALC_SO_MULTIPLE = 3
CALC_APERTURE_INCL = 4
CALC_APERTURE_EXCL = 5
CALC_PROJECTED = 6
# soap_calc_desc.flags
CALC_FLAG_DM_FLOW = 1
CALC_FLAG_ROTATION = 2
CALC_FLAG_INERTIA = 4
CALC_FLAG_STRICT_COPY = 8
CALC_FLAG_PROPERTY_RADIUS = 16
STATS_COLUMNS = 6   # SOAP_STATS_COLUMNS
# SOAP_RADIUS_*: bound-subhalo properties an aperture radius can be a multiple of (internal key -> id)
RADIUS_PROPERTIES = {"HalfMassRadiusTot": 0, "HalfMassRadiusDM": 1, "HalfMassRadiusGas": 2, "HalfMassRadiusStar": 3,
                     "HalfMassRadiusBaryon": 4, "EncloseRadius": 5}

# total / dark matter / gas / stellar inertia tensors, in column order
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

SUBHALO_COLS = {
    "Mtot": (0, 1), "Mgas": (1, 1), "Mdm": (2, 1), "Mstar": (3, 1), "Mbh_dynamical": (4, 1),
    "Ngas": (5, 1), "Ndm": (6, 1), "Nstar": (7, 1), "Nbh": (8, 1),
    "com": (9, 3), "vcom": (12, 3), "Ldm": (15, 3),
    "Vmax_unsoft": (18, 1), "Vmax_soft": (19, 1), "R_vmax_unsoft": (20, 1), "spin_parameter": (21, 1),
    "HalfMassRadiusTot": (22, 1), "HalfMassRadiusDM": (23, 1),
    "veldisp_matrix_dm": (24, 6), "EncloseRadius": (30, 1),
    "com_star": (31, 3), "Lgas": (34, 3), "Lstar": (37, 3), "Lbaryons": (40, 3),
    "veldisp_matrix_gas": (43, 6), "veldisp_matrix_star": (49, 6),
    "HalfMassRadiusGas": (55, 1), "HalfMassRadiusStar": (56, 1), "HalfMassRadiusBaryon": (57, 1),
    "kappa_corot_gas": (58, 1), "kappa_corot_star": (59, 1), "kappa_corot_baryons": (60, 1),
    "DtoTgas": (61, 1), "DtoTstar": (62, 1),
    **{k: (63 + 6 * i, 6) for i, k in enumerate(INERTIA_KEYS)},
}  # fmt: skip
SO_COLS = {
    "r": (0, 1), "Mtot": (1, 1), "Ngas": (2, 1), "Ndm": (3, 1), "Nstar": (4, 1), "Nbh": (5, 1),
    "com": (6, 3), "vcom": (9, 3), "Vmax_soft": (12, 1), "R_vmax_soft": (13, 1),
    "Mfrac_satellites": (14, 1), "Mfrac_external": (15, 1), "Mgas": (16, 1), "Mdm": (17, 1),
    "Ldm": (18, 3), "Mstar": (21, 1), "Mbh_dynamical": (22, 1), "spin_parameter": (23, 1),
    "concentration_unsoft": (24, 1), "concentration_soft": (25, 1),
    "com_gas": (26, 3), "com_star": (29, 3), "vcom_gas": (32, 3), "vcom_star": (35, 3),
    "Lgas": (38, 3), "Lstar": (41, 3), "Lbaryons": (44, 3),
    "Nnu": (47, 1), "Mnu": (48, 1), "MnuNS": (49, 1),
    "DarkMatterMassFlowRate": (50, 6),
    "DtoTgas": (56, 1), "DtoTstar": (57, 1),
    **{k: (58 + 6 * i, 6) for i, k in enumerate(INERTIA_KEYS)},
}  # fmt: skip
APERTURE_COLS = {
    "Mtot": (0, 1), "Mgas": (1, 1), "Mdm": (2, 1), "Mstar": (3, 1), "Mbh_dynamical": (4, 1),
    "Ngas": (5, 1), "Ndm": (6, 1), "Nstar": (7, 1), "Nbh": (8, 1),
    "com": (9, 3), "vcom": (12, 3), "Ldm": (15, 3), "veldisp_matrix_dm": (18, 6),
    "HalfMassRadiusDM": (24, 1), "Vmax_soft": (25, 1), "R_vmax_soft": (26, 1),
    "com_star": (27, 3), "vcom_star": (30, 3), "Lgas": (33, 3), "Lstar": (36, 3), "Lbaryons": (39, 3),
    "veldisp_matrix_gas": (42, 6), "veldisp_matrix_star": (48, 6),
    "HalfMassRadiusGas": (54, 1), "HalfMassRadiusStar": (55, 1), "HalfMassRadiusBaryon": (56, 1),
    "kappa_corot_gas": (57, 1), "kappa_corot_star": (58, 1), "kappa_corot_baryons": (59, 1),
    "DtoTgas": (60, 1), "DtoTstar": (61, 1),
}  # fmt: skip
PROJECTED_COLS = {
    "Mtot": (0, 1), "Mgas": (1, 1), "Mdm": (2, 1), "Mstar": (3, 1), "Mbh_dynamical": (4, 1),
    "Ngas": (5, 1), "Ndm": (6, 1), "Nstar": (7, 1), "Nbh": (8, 1),
    "com": (9, 3), "vcom": (12, 3), "HalfMassRadiusDM": (15, 1), "proj_veldisp_dm": (16, 1),
    "com_star": (17, 3), "proj_veldisp_gas": (20, 1), "proj_veldisp_star": (21, 1),
    "HalfMassRadiusGas": (22, 1), "HalfMassRadiusStar": (23, 1), "HalfMassRadiusBaryon": (24, 1),
}  # fmt: skip
PROJECTED_NCOLS = 25

NCOLS = {
    CALC_SUBHALO: 159,
    CALC_SO_DENSITY: 154,
    CALC_SO_PHYSICAL: 154,
    CALC_SO_MULTIPLE: 154,
    CALC_APERTURE_INCL: 62,
    CALC_APERTURE_EXCL: 62,
    CALC_PROJECTED: 3 * PROJECTED_NCOLS,
}


def columns_of(kind):
    if kind == CALC_SUBHALO:
        return SUBHALO_COLS
    if kind in (CALC_SO_DENSITY, CALC_SO_PHYSICAL, CALC_SO_MULTIPLE):
        return SO_COLS
    if kind in (CALC_APERTURE_INCL, CALC_APERTURE_EXCL):
        return APERTURE_COLS
    if kind == CALC_PROJECTED:
        return PROJECTED_COLS
    raise KeyError(kind)


# properties that a dark-matter-only run carries as a column of zeros without any engine column behind them
# (ExSituFraction: `Nstar == 0` gives None, aperture_properties.py:764-769); with stars they would need
# datasets outside this path, so they are only emitted in DMO runs
VIRTUAL_ZERO_KEYS = ("ExSituFraction",)


# 2D inertia tensors of projected apertures: after the three projection blocks, 36 columns per projection
PROJECTED_INERTIA_KEYS = tuple(
    f"Projected{comp}InertiaTensor{suffix}"
    for comp in ("Total", "Gas", "Stellar")
    for suffix in ("", "Reduced", "Noniterative", "ReducedNoniterative")
)
PROJECTED_INERTIA_PER_AXIS = 3 * len(PROJECTED_INERTIA_KEYS)


def projected_columns(key, iproj):
    """(first column, number of columns) of `key` for projection `iproj` inside a projected aperture's block."""
    if key in PROJECTED_INERTIA_KEYS:
        return 3 * PROJECTED_NCOLS + iproj * PROJECTED_INERTIA_PER_AXIS + 3 * PROJECTED_INERTIA_KEYS.index(key), 3
    c0, nc = PROJECTED_COLS[key]
    return iproj * PROJECTED_NCOLS + c0, nc


def desc_columns(desc) -> int:
    """Columns a calculation occupies in one run (mirror of soap_engine_desc_columns): the full width of its
    kind, minus the 96 inertia tensor columns that end the subhalo / SO blocks when they are not wanted."""
    nc = NCOLS[desc.kind]
    if desc.kind <= CALC_SO_MULTIPLE and not (desc.flags & CALC_FLAG_INERTIA):
        nc -= 6 * len(INERTIA_KEYS)
    if desc.kind == CALC_PROJECTED and (desc.flags & CALC_FLAG_INERTIA):
        nc += 3 * PROJECTED_INERTIA_PER_AXIS
    return nc
