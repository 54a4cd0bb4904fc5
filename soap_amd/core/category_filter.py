"""
Category filter: which calculations run for a halo, decided from the particle counts of
the bound subhalo (SOAP/core/category_filter.py:30-106).
"""

from __future__ import annotations

from typing import Dict

from ..engine import SLOT_OF_PTYPE

_COUNT_SLOT = {
    "BoundSubhalo/NumberOfGasParticles": 0,
    "BoundSubhalo/NumberOfDarkMatterParticles": 1,
    "BoundSubhalo/NumberOfStarParticles": 2,
    "BoundSubhalo/NumberOfBlackHoleParticles": 3,
}


def default_filters(limits: Dict[str, int] | None = None) -> Dict:
    """Same categories as the reference's parameter files / test generator
    (tests/dummy_halo_generator.py:600-634)."""
    limits = limits or {}
    n = "BoundSubhalo/NumberOf"
    return {
        "general": {"limit": limits.get("general", 100),
                    "properties": [n + "DarkMatterParticles", n + "GasParticles",
                                   n + "StarParticles", n + "BlackHoleParticles"],
                    "combine_properties": "sum"},
        "dm": {"limit": limits.get("dm", 100), "properties": [n + "DarkMatterParticles"]},
        "gas": {"limit": limits.get("gas", 100), "properties": [n + "GasParticles"]},
        "star": {"limit": limits.get("star", 100), "properties": [n + "StarParticles"]},
        "baryon": {"limit": limits.get("baryon", 100),
                   "properties": [n + "GasParticles", n + "StarParticles"],
                   "combine_properties": "sum"},
    }  # fmt: skip


class CategoryFilter:
    def __init__(self, filters: Dict | None = None, dmo: bool = False):
        self.filters = default_filters() if filters is None else filters
        self.dmo = dmo

    def get_do_calculation(self, halo_result: Dict, precomputed_properties: Dict | None = None):
        pre = dict(precomputed_properties or {})
        do_calculation = {"basic": True}
        if self.dmo:
            pre["BoundSubhalo/NumberOfGasParticles"] = 0
            pre["BoundSubhalo/NumberOfStarParticles"] = 0
            pre["BoundSubhalo/NumberOfBlackHoleParticles"] = 0
        for name, info in self.filters.items():
            if len(info["properties"]) == 1 or info.get("combine_properties") == "sum":
                v = 0
                for prop in info["properties"]:
                    if prop in pre:
                        v += pre[prop]
                    else:
                        val = halo_result[prop][0]
                        val = getattr(val, "value", val)
                        v += int(val.reshape(-1)[0]) if hasattr(val, "reshape") else int(val)
                do_calculation[name] = v >= info["limit"]
            else:
                raise NotImplementedError(f"Invalid combine_properties function for filter {name}")
        return do_calculation

    def get_filter_metadata(self, halo_filter: str) -> Dict:
        if halo_filter == "basic":
            return {"Masked": False}
        info = self.filters[halo_filter]
        return {"Masked": True, "MaskThreshold": info["limit"], "MaskDatasets": info["properties"]}

    def engine_filter(self, halo_filter: str):
        """(limit, type-slot bit mask) the engine evaluates on the device."""
        if halo_filter == "basic":
            return 0, 0
        info = self.filters[halo_filter]
        mask = 0
        for prop in info["properties"]:
            slot = _COUNT_SLOT[prop]
            if self.dmo and slot != 1:
                continue  # DMO: gas/star/BH counts are forced to zero (category_filter.py:87-91)
            mask |= 1 << slot
        return int(info["limit"]), mask
