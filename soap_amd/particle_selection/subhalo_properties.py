"""
BoundSubhalo: all particles with GroupNr_bound == halo index
(SOAP/particle_selection/subhalo_properties.py:128-164, :2345-2740).
Must be the first entry of halo_prop_list: its particle counts feed every category filter
and its EncloseRadius feeds the aperture skip/copy rule
(SOAP/compute_halo_properties.py:232-247).
"""

from __future__ import annotations

from .. import layout
from .halo_properties import HaloProperty


class SubhaloProperties(HaloProperty):
    base_halo_type = "SubhaloProperties"
    kind = layout.CALC_SUBHALO
    # subhalo_properties.py:2557-2567
    _PARTICLE_DATASETS = {
        "PartType0": ("Coordinates", "Masses", "Velocities", "GroupNr_bound"),
        "PartType1": ("Coordinates", "Masses", "Velocities", "GroupNr_bound"),
        "PartType4": ("Coordinates", "Masses", "Velocities", "GroupNr_bound"),
        "PartType5": ("Coordinates", "DynamicalMasses", "Velocities", "GroupNr_bound"),
    }
    property_names = [
        "Mtot", "Mgas", "Mdm", "Mstar", "Mbh_dynamical", "Ngas", "Ndm", "Nstar", "Nbh",
        "com", "vcom", "Ldm", "Vmax_unsoft", "Vmax_soft", "R_vmax_unsoft", "spin_parameter",
        "HalfMassRadiusTot", "HalfMassRadiusDM", "veldisp_matrix_dm", "EncloseRadius",
        "com_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas", "veldisp_matrix_star",
        "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
        "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar", "ExSituFraction",
        *layout.INERTIA_KEYS,
    ]  # fmt: skip

    def __init__(self, cellgrid, parameters=None, recently_heated_gas_filter=None,
                 stellar_age_calculator=None, category_filter=None):  # fmt: skip
        super().__init__(cellgrid)
        self.category_filter = category_filter
        self.property_filters = _property_filters(parameters, self)
        self.name = "bound_subhalo"
        self.group_name = "BoundSubhalo"
        self.mask_metadata = {"Masked": False}
        self.halo_filter = "basic"
        self.physical_radius_mpc = 0.0

    def calc_desc(self, halo_prop_list, position):
        return self._base_desc()


def _property_filters(parameters, prop_calc):
    """parameters.get_property_filters(base_halo_type, names) of the reference
    (core/parameter_file.py:85-154); None = everything enabled with filter "basic"."""
    from ..property_table import full_property_list

    names = [full_property_list[k].name for k in prop_calc.property_names]
    if parameters is None:
        return {n: "basic" for n in names}
    return parameters.get_property_filters(prop_calc.base_halo_type, names)
