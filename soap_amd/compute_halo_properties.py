"""
Assembly of `halo_prop_list` in the reference's fixed order and with its default
variations (SOAP/compute_halo_properties.py:232-505): bound subhalo first, SO definitions,
SO radius multiples, spherical apertures sorted by radius, projected apertures.

Only the list construction is mirrored here -- the reference's driver (MPI set-up, snapshot
I/O, chunk task farm, catalogue writer) is the *caller* of the hot path and out of scope.
"""

from __future__ import annotations

from typing import Dict, List

from .particle_selection import (SO_properties, aperture_properties, projected_aperture_properties,
                                 subhalo_properties)  # fmt: skip

# compute_halo_properties.py:249-262
DEFAULT_SO_VARIATIONS = {
    "200_mean": {"value": 200.0, "type": "mean"},
    "50_crit": {"value": 50.0, "type": "crit"},
    "100_crit": {"value": 100.0, "type": "crit"},
    "200_crit": {"value": 200.0, "type": "crit"},
    "500_crit": {"value": 500.0, "type": "crit"},
    "1000_crit": {"value": 1000.0, "type": "crit"},
    "2500_crit": {"value": 2500.0, "type": "crit"},
    "BN98": {"value": 0.0, "type": "BN98"},
    "5xR500_crit": {"value": 500.0, "type": "crit", "radius_multiple": 5.0},
}


def aperture_variations(radii_kpc=(30.0, 50.0, 100.0)) -> Dict:
    """Inclusive + exclusive spheres of the given radii, named as in
    compute_halo_properties.py:317-337."""
    out = {}
    for r in radii_kpc:
        out[f"inclusive_{r:.0f}_kpc"] = {"radius_in_kpc": float(r), "inclusive": True}
    for r in radii_kpc:
        out[f"exclusive_{r:.0f}_kpc"] = {"radius_in_kpc": float(r), "inclusive": False}
    return out


def projected_variations(radii_kpc=(10.0, 30.0, 50.0, 100.0)) -> Dict:
    """compute_halo_properties.py:442-450"""
    return {f"{r:.0f}_kpc": {"radius_in_kpc": float(r)} for r in radii_kpc}


def build_halo_prop_list(cellgrid, category_filter, SO_variations=None, aperture_vars=None,
                         parameters=None, classes=None, projected_vars=None) -> List:  # fmt: skip
    """`classes` lets the parity tests build the same list out of the oracle's classes."""
    if classes is None:
        classes = dict(
            subhalo=lambda: subhalo_properties.SubhaloProperties(
                cellgrid, parameters, None, None, category_filter),
            so=lambda filt, val, typ: SO_properties.SOProperties(
                cellgrid, parameters, None, category_filter, filt, val, typ),
            so_core_excised=lambda filt, val, typ, frac: SO_properties.CoreExcisedSOProperties(
                cellgrid, parameters, None, category_filter, filt, val, typ, core_excision_fraction=frac),
            so_multiple=lambda filt, val, mult, typ: SO_properties.RadiusMultipleSOProperties(
                cellgrid, parameters, None, category_filter, filt, val, mult, typ),
            inclusive=lambda filt, r, radii: aperture_properties.InclusiveSphereProperties(
                cellgrid, parameters, r, None, None, None, None, category_filter, filt, radii),
            exclusive=lambda filt, r, radii: aperture_properties.ExclusiveSphereProperties(
                cellgrid, parameters, r, None, None, None, None, category_filter, filt, radii),
            inclusive_property=lambda filt, prop: aperture_properties.InclusiveSphereProperties(
                cellgrid, parameters, None, prop, None, None, None, category_filter, filt, []),
            exclusive_property=lambda filt, prop: aperture_properties.ExclusiveSphereProperties(
                cellgrid, parameters, None, prop, None, None, None, category_filter, filt, []),
            projected=lambda filt, r, radii: projected_aperture_properties.ProjectedApertureProperties(
                cellgrid, parameters, r, None, category_filter, filt, radii),
        )  # fmt: skip
    SO_variations = DEFAULT_SO_VARIATIONS if SO_variations is None else SO_variations
    aperture_vars = {} if aperture_vars is None else aperture_vars
    halo_prop_list = [classes["subhalo"]()]
    # non radius-multiples first (compute_halo_properties.py:263-297)
    for v in SO_variations.values():
        if v.get("radius_multiple", 0.0) > 0.0:
            continue
        if "core_excision_fraction" in v and "so_core_excised" in classes:   # compute_halo_properties.py:271-285
            halo_prop_list.append(classes["so_core_excised"](v.get("filter", "basic"), v["value"], v["type"], v["core_excision_fraction"]))
        else:
            halo_prop_list.append(classes["so"](v.get("filter", "basic"), v["value"], v["type"]))
    for v in SO_variations.values():
        if v.get("radius_multiple", 0.0) > 0.0:
            halo_prop_list.append(classes["so_multiple"](
                v.get("filter", "basic"), v["value"], v["radius_multiple"], v["type"]))  # fmt: skip
    # apertures sorted by radius (compute_halo_properties.py:339-400)
    aperture_vars = dict(sorted(aperture_vars.items(), key=lambda x: x[1].get("radius_in_kpc", 0)))
    inclusive_radii_kpc = [v["radius_in_kpc"] for v in aperture_vars.values() if v["inclusive"] and "radius_in_kpc" in v]
    exclusive_radii_kpc = [v["radius_in_kpc"] for v in aperture_vars.values() if not v["inclusive"] and "radius_in_kpc" in v]
    for v in aperture_vars.values():
        if "radius_in_kpc" not in v:
            continue
        assert "property" not in v and "radius_multiple" not in v
        if v["inclusive"]:
            radii_kpc = [v["radius_in_kpc"]]
            if v.get("skip_gt_enclose_radius", False):
                radii_kpc = inclusive_radii_kpc
            halo_prop_list.append(classes["inclusive"](v.get("filter", "basic"), v["radius_in_kpc"], radii_kpc))
        else:
            halo_prop_list.append(classes["exclusive"](v.get("filter", "basic"), v["radius_in_kpc"], exclusive_radii_kpc))
    # apertures whose radius is a multiple of a bound-subhalo property (compute_halo_properties.py:403-440)
    for v in aperture_vars.values():
        if "radius_in_kpc" in v:
            continue
        assert "property" in v
        radius_multiple = v.get("radius_multiple", 1)
        assert int(radius_multiple) == radius_multiple   # swiftsimio could not handle the group names otherwise
        cls = classes["inclusive_property" if v["inclusive"] else "exclusive_property"]
        halo_prop_list.append(cls(v.get("filter", "basic"), (v["property"], radius_multiple)))
    # projected apertures sorted by radius (compute_halo_properties.py:451-489)
    projected_vars = {} if projected_vars is None else projected_vars
    projected_vars = dict(sorted(projected_vars.items(), key=lambda x: x[1].get("radius_in_kpc", 0)))
    projected_radii_kpc = [v["radius_in_kpc"] for v in projected_vars.values()]
    for v in projected_vars.values():
        halo_prop_list.append(classes["projected"](v.get("filter", "basic"), v["radius_in_kpc"], projected_radii_kpc))
    return halo_prop_list
