"""
The calculation lists and filters of tests/golden/control_flow.npz, in ONE place: the generator
(tests/golden/make_control_flow_golden.py) builds the REFERENCE's classes from these dictionaries, the tests build the
oracle's and the engine's from the same ones (through compute_halo_properties.build_halo_prop_list, which orders a list
the way the reference's main program does).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "control_flow.npz")
# the same cases at a = 0.769 with quantities that carry units (comoving coordinates, physical densities ...): every
# comoving <-> physical conversion is the reference's own (tests/golden/_shims_units)
GOLDEN_UNITS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "control_flow_units.npz")
# the same cases at a = 1 with Velocities and Masses in float32, as SWIFT snapshots hold them (make_control_flow_golden.py --f32): the
# reference casts the cumulative mass of the SO profile back to float32 (SO_properties.py:400-402) and sums in float32
GOLDEN_F32 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "control_flow_f32.npz")

# category filters (core/category_filter.py: limit on the summed particle numbers of the bound subhalo)
FILTERS = {
    "general": {"limit": 100, "properties": ["BoundSubhalo/NumberOfDarkMatterParticles"], "combine_properties": "sum"},
    "dm": {"limit": 100, "properties": ["BoundSubhalo/NumberOfDarkMatterParticles"], "combine_properties": "sum"},
    "baryon": {"limit": 60, "properties": ["BoundSubhalo/NumberOfGasParticles", "BoundSubhalo/NumberOfStarParticles"],
               "combine_properties": "sum"},
}  # fmt: skip

# name: (box, dmo, strict_halo_copy, skip_gt_enclose_radius of the inclusive spheres)
VARIANTS = {
    "default": ("dmo", True, False, False),
    "strict_skip": ("dmo", True, True, True),
    "hydro": ("hydro", False, False, False),
    "hydro_strict": ("hydro", False, True, True),
}
MESH_RESOLUTION = 6
RADII_KPC = (100.0, 1000.0, 3000.0)
PROJECTED_RADII_KPC = (1000.0, 3000.0)


def variations(dmo, skip_incl):
    if dmo:
        so = {
            "200_crit": {"type": "crit", "value": 200.0}, "50_crit": {"type": "crit", "value": 50.0},
            "2500_crit": {"type": "crit", "value": 2500.0, "filter": "general"}, "200_mean": {"type": "mean", "value": 200.0},
            "BN98": {"type": "BN98", "value": 0.0}, "500_crit": {"type": "crit", "value": 500.0},
            "100_kpc": {"type": "physical", "value": 100.0},
            "5xR_500_crit": {"type": "crit", "value": 500.0, "radius_multiple": 5.0},
        }  # fmt: skip
    else:
        so = {"200_crit": {"type": "crit", "value": 200.0}, "500_crit": {"type": "crit", "value": 500.0, "filter": "baryon"}}
    ap = {}
    for r in RADII_KPC:
        ap[f"inclusive_{r:.0f}_kpc"] = {"radius_in_kpc": r, "inclusive": True, "skip_gt_enclose_radius": skip_incl}
        ap[f"exclusive_{r:.0f}_kpc"] = {"radius_in_kpc": r, "inclusive": False,
                                        "filter": ("dm" if dmo else "baryon") if r == RADII_KPC[-1] else "basic"}  # fmt: skip
    # apertures whose radius is a multiple of a bound-subhalo property (compute_halo_properties.py:403-440)
    if dmo:
        ap["exclusive_3xHalfMassRadiusDarkMatter"] = {"property": "BoundSubhalo/HalfMassRadiusDarkMatter", "radius_multiple": 3,
                                                      "inclusive": False}
        ap["inclusive_2xEncloseRadius"] = {"property": "BoundSubhalo/EncloseRadius", "radius_multiple": 2, "inclusive": True}
    else:
        ap["exclusive_2xHalfMassRadiusStars"] = {"property": "BoundSubhalo/HalfMassRadiusStars", "radius_multiple": 2,
                                                 "inclusive": False}
        ap["inclusive_HalfMassRadiusTotal"] = {"property": "BoundSubhalo/HalfMassRadiusTotal", "inclusive": True}
        ap["inclusive_4xHalfMassRadiusBaryons"] = {"property": "BoundSubhalo/HalfMassRadiusBaryons", "radius_multiple": 4,
                                                   "inclusive": True, "filter": "baryon"}
        ap["exclusive_HalfMassRadiusGas"] = {"property": "BoundSubhalo/HalfMassRadiusGas", "inclusive": False}
    pj = {f"projected_{r:.0f}_kpc": {"radius_in_kpc": r} for r in PROJECTED_RADII_KPC}
    return so, ap, pj


class Golden:
    """Reader of the packed fixture."""

    def __init__(self, units=False, f32=False):
        self.z = np.load(GOLDEN_F32 if f32 else (GOLDEN_UNITS if units else GOLDEN))

    def scalars(self):
        return {k.split("|")[1]: float(self.z[k]) for k in self.z.files if k.startswith("scalars|")}

    def box(self, bname):
        data = {}
        for k in self.z.files:
            if k.startswith(f"box_{bname}|"):
                _, pt, name = k.split("|")
                data.setdefault(int(pt[len("PartType"):]), {})[name] = self.z[k]
        halos = {k.split("|")[1]: self.z[k] for k in self.z.files if k.startswith(f"halos_{bname}|")}
        return data, halos

    def attempts(self, vname, i):
        """rows: search radius in, read radius in, n_loop of the attempt, done, search radius out"""
        return self.z[f"{vname}|halo{i:02d}|attempts"]

    def done(self, vname, i):
        return bool(self.z[f"{vname}|halo{i:02d}|done"])

    def record(self, vname, i):
        t = f"{vname}|halo{i:02d}"
        keys, sizes, v = self.z[t + "|keys"], self.z[t + "|sizes"], self.z[t + "|values"]
        off = np.concatenate([[0], np.cumsum(sizes)])
        f32 = self.z[t + "|float32"]
        return {str(k): (v[off[j] : off[j + 1]], bool(f32[j])) for j, k in enumerate(keys)}
