"""
Fixed spherical apertures, inclusive (all particles) and exclusive (bound only):
SOAP/particle_selection/aperture_properties.py:3702-4398.
"""

from __future__ import annotations

from .. import layout
from .halo_properties import HaloProperty
from .subhalo_properties import _property_filters


def _aperture_name(radius_kpc):
    if radius_kpc < 1:
        return f"{1000*radius_kpc:.0f}pc"
    return f"{radius_kpc:.0f}kpc"


_AP_DSETS = ("Coordinates", "GroupNr_bound", "Masses", "Velocities")
# aperture_properties.py:3995-4005, projected_aperture_properties.py:1770-1780
APERTURE_PARTICLE_DATASETS = {
    "PartType0": _AP_DSETS, "PartType1": _AP_DSETS, "PartType4": _AP_DSETS,
    "PartType5": ("Coordinates", "DynamicalMasses", "GroupNr_bound", "Velocities"),
}  # fmt: skip


class ApertureProperties(HaloProperty):
    base_halo_type = "ApertureProperties"
    _PARTICLE_DATASETS = APERTURE_PARTICLE_DATASETS
    property_names = [
        "Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar", "Nbh",
        "com", "vcom", "Ldm", "veldisp_matrix_dm", "HalfMassRadiusDM", "Vmax_soft", "R_vmax_soft",
        "com_star", "vcom_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas", "veldisp_matrix_star",
        "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
        "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar", "ExSituFraction",
    ]  # fmt: skip

    def __init__(self, cellgrid, parameters, aperture_physical_radius_kpc, aperture_property,
                 recently_heated_gas_filter, stellar_age_calculator, cold_dense_gas_filter,
                 category_filter, halo_filter, inclusive, all_radii_kpc):  # fmt: skip
        super().__init__(cellgrid)
        if (aperture_physical_radius_kpc is None) == (aperture_property is None):
            raise ValueError("an aperture needs either a physical radius or a defining property")
        self.category_filter = category_filter
        self.halo_filter = halo_filter
        self.property_filters = _property_filters(parameters, self)
        self.all_radii_kpc = list(all_radii_kpc)
        # parameters.strict_halo_copy() (aperture_properties.py:3953, :4117-4120): a copied sphere leaves the
        # properties with a direct dependence on the aperture size at zero.  In ApertureProperties.property_names
        # those are the four StellarInertiaTensor* entries (:3716-3873), none of which this engine emits: the
        # switch is read and kept, and changes no column of a spherical aperture.
        self.strict_halo_copy = bool(parameters.strict_halo_copy()) if hasattr(parameters, "strict_halo_copy") else False
        self.aperture_physical_radius_kpc = aperture_physical_radius_kpc
        self.aperture_property = aperture_property
        self.inclusive = inclusive
        if aperture_physical_radius_kpc is not None:
            assert aperture_physical_radius_kpc >= 0.001
            self.physical_radius_mpc = 0.001 * aperture_physical_radius_kpc
            aperture_name = _aperture_name(aperture_physical_radius_kpc)
        else:
            # radius = multiple x a property of the bound subhalo (aperture_properties.py:3973-3981)
            group, prop = aperture_property[0].split("/")[0], aperture_property[0].split("/")[-1]
            assert group == "BoundSubhalo"
            multiplier = aperture_property[1]
            aperture_name = prop if multiplier == 1 else f"{int(multiplier)}x{prop}"
            # "used to guess the initial load region for each particle" (:3979-3981)
            self.physical_radius_mpc = 0
            from ..property_table import full_property_list

            by_name = {full_property_list[k].name: k for k in layout.RADIUS_PROPERTIES}
            if prop not in by_name:
                raise NotImplementedError(
                    f"aperture radius from {aperture_property[0]}: the engine knows the half-mass radii and the enclose "
                    f"radius of the bound subhalo ({sorted(by_name)})")
            self._radius_property = layout.RADIUS_PROPERTIES[by_name[prop]]
            self._radius_property_key = by_name[prop]
        if inclusive:
            self.name = f"inclusive_sphere_{aperture_name}"
            self.group_name = f"InclusiveSphere/{aperture_name}"
            self.kind = layout.CALC_APERTURE_INCL
        else:
            self.name = f"exclusive_sphere_{aperture_name}"
            self.group_name = f"ExclusiveSphere/{aperture_name}"
            self.kind = layout.CALC_APERTURE_EXCL
        self.mask_metadata = (
            category_filter.get_filter_metadata(halo_filter) if category_filter else {"Masked": False}
        )

    def calc_desc(self, halo_prop_list, position):
        d = self._base_desc()
        if self.aperture_property is not None:
            # :4131-4138: the property has to be in halo_result, i.e. enabled for the bound subhalo
            sub = halo_prop_list[0]
            if self._radius_property_key not in sub.enabled_properties():
                raise RuntimeError(f"{self.aperture_property[0]} must be enabled in the parameter file if you want to "
                                   "use it to define an aperture")
            d.radius = float(self.aperture_property[1])
            d.parent = self._radius_property
            d.flags |= layout.CALC_FLAG_PROPERTY_RADIUS
            d.physical_radius_mpc = 0.0
            return d
        d.radius = self.cellgrid.kpc_to_comoving(self.aperture_physical_radius_kpc)
        # skip/copy rule (aperture_properties.py:4082-4123)
        i_radius = self.all_radii_kpc.index(self.aperture_physical_radius_kpc)
        if i_radius != 0:
            r_prev = self.all_radii_kpc[i_radius - 1]
            prefix = "InclusiveSphere" if self.inclusive else "ExclusiveSphere"
            prev_group = f"{prefix}/{_aperture_name(r_prev)}"
            for i, hp in enumerate(halo_prop_list[:position]):
                if getattr(hp, "group_name", None) == prev_group:
                    d.parent = i
            if d.parent < 0:
                raise RuntimeError(f"{prev_group} must be computed before {self.group_name}")
            d.prev_radius = self.cellgrid.kpc_to_comoving(r_prev)
        return d


class ExclusiveSphereProperties(ApertureProperties):
    def __init__(self, cellgrid, parameters, physical_radius_kpc, aperture_property,
                 recently_heated_gas_filter, stellar_age_calculator, cold_dense_gas_filter,
                 category_filter, halo_filter, all_radii_kpc):  # fmt: skip
        super().__init__(cellgrid, parameters, physical_radius_kpc, aperture_property,
                         recently_heated_gas_filter, stellar_age_calculator,
                         cold_dense_gas_filter, category_filter, halo_filter, False,
                         all_radii_kpc)  # fmt: skip


class InclusiveSphereProperties(ApertureProperties):
    def __init__(self, cellgrid, parameters, physical_radius_kpc, aperture_property,
                 recently_heated_gas_filter, stellar_age_calculator, cold_dense_gas_filter,
                 category_filter, halo_filter, all_radii_kpc):  # fmt: skip
        super().__init__(cellgrid, parameters, physical_radius_kpc, aperture_property,
                         recently_heated_gas_filter, stellar_age_calculator,
                         cold_dense_gas_filter, category_filter, halo_filter, True,
                         all_radii_kpc)  # fmt: skip
