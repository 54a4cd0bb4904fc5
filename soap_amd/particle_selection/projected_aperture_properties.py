"""
Projected apertures: bound particles inside a fixed 2D radius, for projections along x, y, z
(SOAP/particle_selection/projected_aperture_properties.py:1580-2002).
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from .. import layout
from ..property_table import full_property_list, stored_value
from .aperture_properties import APERTURE_PARTICLE_DATASETS, _aperture_name
from .halo_properties import HaloProperty
from .subhalo_properties import _property_filters


class ProjectedApertureProperties(HaloProperty):
    base_halo_type = "ProjectedApertureProperties"
    kind = layout.CALC_PROJECTED
    _PARTICLE_DATASETS = APERTURE_PARTICLE_DATASETS
    property_names = ["Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar", "Nbh",
                      "com", "vcom", "HalfMassRadiusDM", "proj_veldisp_dm",
                      "com_star", "proj_veldisp_gas", "proj_veldisp_star",
                      "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
                      *layout.PROJECTED_INERTIA_KEYS]  # fmt: skip

    def __init__(self, cellgrid, parameters, aperture_physical_radius_kpc, aperture_property,
                 category_filter, halo_filter, all_radii_kpc):  # fmt: skip
        super().__init__(cellgrid)
        if aperture_physical_radius_kpc is None or aperture_property is not None:
            raise NotImplementedError("projected apertures defined by another property are not built yet")
        self.category_filter = category_filter
        self.halo_filter = halo_filter
        self.property_filters = _property_filters(parameters, self)
        self.all_radii_kpc = list(all_radii_kpc)
        # parameters.strict_halo_copy() (projected_aperture_properties.py:1742, :1880-1883): a copied aperture leaves
        # the iterative 2D inertia tensors (Projected{Total,Gas,Stellar}InertiaTensor{,Reduced}: `True` in
        # property_names, :1596-1676) at zero; the engine gets the switch as SOAP_CALC_FLAG_STRICT_COPY
        self.strict_halo_copy = bool(parameters.strict_halo_copy()) if hasattr(parameters, "strict_halo_copy") else False
        self.aperture_physical_radius_kpc = aperture_physical_radius_kpc
        self.physical_radius_mpc = 0.001 * aperture_physical_radius_kpc
        aperture_name = _aperture_name(aperture_physical_radius_kpc)
        self.name = f"projected_aperture_{aperture_name}"
        self.group_name = f"ProjectedAperture/{aperture_name}"
        self.mask_metadata = (
            category_filter.get_filter_metadata(halo_filter) if category_filter else {"Masked": False}
        )

    def calc_desc(self, halo_prop_list, position):
        d = self._base_desc()
        d.radius = self.cellgrid.kpc_to_comoving(self.aperture_physical_radius_kpc)
        i_radius = self.all_radii_kpc.index(self.aperture_physical_radius_kpc)
        if i_radius != 0:
            r_prev = self.all_radii_kpc[i_radius - 1]
            prev_group = f"ProjectedAperture/{_aperture_name(r_prev)}"
            for i, hp in enumerate(halo_prop_list[:position]):
                if getattr(hp, "group_name", None) == prev_group:
                    d.parent = i
            if d.parent < 0:
                raise RuntimeError(f"{prev_group} must be computed before {self.group_name}")
            d.prev_radius = self.cellgrid.kpc_to_comoving(r_prev)
        if self.strict_halo_copy:
            d.flags |= layout.CALC_FLAG_STRICT_COPY
        return d

    def unpack(self, block: np.ndarray, halo_result: Dict, do_calculation=None):
        for iproj, projname in enumerate(("projx", "projy", "projz")):
            for key in self.enabled_properties():
                prop = full_property_list[key]
                c0, nc = layout.projected_columns(key, iproj)
                raw = block[c0 : c0 + nc].copy()
                filt = self.property_filters.get(prop.name, "basic")
                if do_calculation is not None and not do_calculation.get(filt, True):
                    raw[:] = 0.0
                halo_result[f"{self.group_name}/{projname}/{prop.name}"] = (
                    stored_value(key, raw, self.a), prop.description, prop.output_physical,
                    prop.a_scale_exponent, raw,
                )  # fmt: skip
