"""
Spherical overdensities: SOProperties, CoreExcisedSOProperties and RadiusMultipleSOProperties with the reference's
constructor signatures, names and density bookkeeping
(SOAP/particle_selection/SO_properties.py:3215-3567, :3797-3926).
"""

from __future__ import annotations

from .. import layout
from .halo_properties import HaloProperty
from .subhalo_properties import _property_filters


_SO_DSETS = ("Coordinates", "FOFGroupIDs", "GroupNr_bound", "Masses", "Velocities")


class SOProperties(HaloProperty):
    base_halo_type = "SOProperties"
    # SO_properties.py:3520-3550 (the FOF ids feed MassFractionSatellites / External, PartType6 the neutrino masses)
    _PARTICLE_DATASETS = {
        "PartType0": _SO_DSETS, "PartType1": _SO_DSETS, "PartType4": _SO_DSETS,
        "PartType5": ("Coordinates", "DynamicalMasses", "FOFGroupIDs", "GroupNr_bound", "Velocities"),
        "PartType6": ("Coordinates", "Masses", "Weights"),
    }  # fmt: skip
    property_names = [
        "r", "Mtot", "Ngas", "Ndm", "Nstar", "Nbh", "com", "vcom", "Vmax_soft", "R_vmax_soft",
        "Mfrac_satellites", "Mfrac_external", "Mgas", "Mdm", "Ldm", "Mstar", "Mbh_dynamical",
        "spin_parameter", "concentration_unsoft", "concentration_soft",
        "com_gas", "com_star", "vcom_gas", "vcom_star", "Lgas", "Lstar", "Lbaryons",
        "Nnu", "Mnu", "MnuNS", "DarkMatterMassFlowRate", "DtoTgas", "DtoTstar",
        *layout.INERTIA_KEYS,
    ]  # fmt: skip

    def __init__(self, cellgrid, parameters, recently_heated_gas_filter, category_filter,
                 halo_filter, SOval, type="mean"):  # fmt: skip
        super().__init__(cellgrid)
        if type not in ["mean", "crit", "physical", "BN98"]:
            raise AttributeError(f"Unknown SO type: {type}!")
        self.type = type
        self.category_filter = category_filter
        self.halo_filter = halo_filter
        self.property_filters = _property_filters(parameters, self)
        # SO_properties.py:3453-3471
        self.mean_density_multiple = 1000.0
        self.critical_density_multiple = 1000.0
        self.physical_radius_mpc = 0.0
        self.virial_definition = False
        if type == "mean":
            self.mean_density_multiple = SOval
            if SOval == 200:
                self.virial_definition = True
        elif type == "crit":
            self.critical_density_multiple = SOval
            if SOval == 200:
                self.virial_definition = True
        elif type == "BN98":
            self.critical_density_multiple = cellgrid.virBN98
            self.virial_definition = True
        elif type == "physical":
            self.physical_radius_mpc = 0.001 * SOval
        if type in ["mean", "crit"]:
            self.name = f"SO_{SOval:.0f}_{type}"
        elif type == "physical":
            self.name = f"SO_{SOval:.0f}_kpc"
        elif type == "BN98":
            self.name = "SO_BN98"
        # SO_properties.py:3490-3509 -- physical density, converted once to comoving
        if self.type == "crit":
            rho = self.critical_density_multiple * self.critical_density
            self.SO_name = f"{self.critical_density_multiple:.0f}_crit"
            self.label = f"within which the density is {self.critical_density_multiple:.0f} times the critical value"
        elif self.type == "mean":
            rho = self.mean_density_multiple * self.mean_density
            self.SO_name = f"{self.mean_density_multiple:.0f}_mean"
            self.label = f"within which the density is {self.mean_density_multiple:.0f} times the mean value"
        elif self.type == "physical":
            rho = 0.0
            self.SO_name = f"{1000. * self.physical_radius_mpc:.0f}_kpc"
            self.label = f"with a radius of {1000. * self.physical_radius_mpc:.0f} kpc"
        else:
            rho = self.critical_density_multiple * self.critical_density
            self.SO_name = "BN98"
            self.label = f"within which the density is {self.critical_density_multiple:.2f} times the critical value"
        self.reference_density = cellgrid.density_to_comoving(rho)
        self.group_name = f"SO/{self.SO_name}"
        self.mask_metadata = (
            category_filter.get_filter_metadata(halo_filter) if category_filter else {"Masked": False}
        )
        self.kind = layout.CALC_SO_PHYSICAL if type == "physical" else layout.CALC_SO_DENSITY

    def calc_desc(self, halo_prop_list, position):
        d = self._base_desc()
        d.reference_density = self.reference_density
        d.radius = self.cellgrid.mpc_to_comoving(self.physical_radius_mpc)
        d.virial_definition = 1 if self.virial_definition else 0
        # DarkMatterMassFlowRate is only defined for the virial definitions (SO_properties.py:2923-2924); when it
        # is wanted the calculation needs the sphere out to 1.05 R_SO (:2849-2850)
        if self.virial_definition and self.kind == layout.CALC_SO_DENSITY and "DarkMatterMassFlowRate" in self.enabled_properties():
            d.flags |= layout.CALC_FLAG_DM_FLOW
        return d


class CoreExcisedSOProperties(SOProperties):
    """
    SOProperties plus the eighteen core-excised temperatures and X-ray luminosities (SO_properties.py:3745-3795): what the
    reference builds for an SO variation that carries `core_excision_fraction` (compute_halo_properties.py:271-285).  Every
    one of the extra properties is a sub-grid quantity outside this engine's set, so the
    class emits exactly what SOProperties emits -- same name, same group, same descriptor -- and a parameter file that lists
    such a variation goes through `build_halo_prop_list` unchanged.
    """

    def __init__(self, cellgrid, parameters, recently_heated_gas_filter, category_filter,
                 halo_filter, SOval, type="mean", core_excision_fraction=None):  # fmt: skip
        self.core_excision_fraction = core_excision_fraction
        super().__init__(cellgrid, parameters, recently_heated_gas_filter, category_filter, halo_filter, SOval, type)


class RadiusMultipleSOProperties(SOProperties):
    def __init__(self, cellgrid, parameters, recently_heated_gas_filter, category_filter,
                 halo_filter, SOval, multiple, type="mean"):  # fmt: skip
        if type not in ["mean", "crit"]:
            raise AttributeError(
                "SOs with a radius that is a multiple of another SO radius are only allowed for type mean or crit!"
            )
        super().__init__(cellgrid, parameters, recently_heated_gas_filter, category_filter,
                         halo_filter, 0, "physical")  # fmt: skip
        self.SO_name = f"{multiple:.0f}xR_{SOval:.0f}_{type}"
        self.group_name = f"SO/{self.SO_name}"
        self.label = f"with a radius that is {self.SO_name}"
        self.name = f"SO_{self.SO_name}"
        self.requested_type = type
        self.requested_SOval = SOval
        self.multiple = multiple
        self.kind = layout.CALC_SO_MULTIPLE

    def calc_desc(self, halo_prop_list, position):
        d = self._base_desc()
        parent_group = f"SO/{self.requested_SOval:.0f}_{self.requested_type}"
        parent = -1
        for i, hp in enumerate(halo_prop_list[:position]):
            if getattr(hp, "group_name", None) == parent_group:
                parent = i
        if parent < 0:
            # SO_properties.py:3912-3916
            raise RuntimeError(
                f"Trying to obtain {parent_group}/SORadius, but the corresponding SO radius has not been calculated!"
            )
        d.parent = parent
        d.radius = float(self.multiple)
        d.physical_radius_mpc = 0.0  # set per halo by the engine (SO_properties.py:3917)
        return d
