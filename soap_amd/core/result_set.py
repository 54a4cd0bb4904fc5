"""
Column store of finished halos and the chunk loop that fills it: the counterpart of
`ResultSet` (SOAP/core/result_set.py:74-130), the per-chunk part of `ChunkTask.__call__`
(SOAP/core/chunk_tasks.py:284-340) and the ordering rule of `combine_chunks`
(SOAP/core/combine_chunks.py:38-52).

The reference appends one dictionary per halo and concatenates at the end.  Here the engine hands back a
dense fp64 table, so the columns are cut out of it for all halos at once: one cast per property to the
dtype of `property_table.py`, property-level category filters applied as row masks, datasets named
`"<group>/<PropertyName>"` exactly as the reference names them
(`SO/200_crit/SORadius`, `InclusiveSphere/50kpc/TotalMass`, `ProjectedAperture/30kpc/projx/...`,
`BoundSubhalo/...`, `InputHalos/...`).
"""

from __future__ import annotations

import json
from typing import Dict, List

import numpy as np

from .. import layout
from ..property_table import full_property_list

from .halo_tasks import _INPUT_HALO_COLUMNS   # InputHalos/* datasets every catalogue carries


class ResultSet:
    """Named, typed columns of the halos finished so far (any number of `append_table` calls)."""

    def __init__(self, halo_prop_list: List, engine_desc_columns):
        self.props = list(halo_prop_list)
        ncols = [int(engine_desc_columns(hp.calc_desc(self.props, i))) for i, hp in enumerate(self.props)]
        self.coloff = np.concatenate([[0], np.cumsum(ncols)]).astype(np.int64)
        self._chunks: Dict[str, List[np.ndarray]] = {}
        self.meta: Dict[str, Dict] = {}
        self.n_halo = 0

    # -- filling ------------------------------------------------------------------------------
    def append_table(self, table: np.ndarray, halo_arrays: Dict[str, np.ndarray]) -> None:
        """table: fp64 rows of finished halos (engine layout); halo_arrays: the same halos' catalogue
        columns (index, cofp, is_central, nr_bound_part)."""
        n = table.shape[0]
        if n == 0:
            return
        do_calc = self._row_filters(table)
        for ci, hp in enumerate(self.props):
            block = table[:, self.coloff[ci] : self.coloff[ci + 1]]
            cols = layout.columns_of(hp.kind)
            for iproj, (proj, base) in enumerate(_projections(hp)):
                for key in hp.enabled_properties():
                    prop = full_property_list[key]
                    if key in layout.VIRTUAL_ZERO_KEYS:
                        raw = np.zeros((n, prop.shape))
                    elif hp.kind == layout.CALC_PROJECTED:
                        c0, nc = layout.projected_columns(key, iproj)
                        raw = block[:, c0 : c0 + nc]
                    else:
                        c0, nc = cols[key]
                        raw = block[:, base + c0 : base + c0 + nc]
                    filt = hp.property_filters.get(prop.name, "basic")
                    if ci > 0 and filt != "basic" and filt in do_calc:   # the subhalo itself is never filtered
                        raw = np.where(do_calc[filt][:, None], raw, 0.0)
                    name = f"{hp.group_name}{proj}/{prop.name}"
                    self._put(name, _cast(raw, prop, hp.a), prop.unit, prop.description,
                              prop.output_physical, prop.a_scale_exponent)  # fmt: skip
        for src, (name, dtype, desc, phys, aexp) in _INPUT_HALO_COLUMNS.items():
            unit = "snap_length" if src == "cofp" else "dimensionless"
            self._put(name, np.asarray(halo_arrays[src]).astype(dtype), unit, desc, phys, aexp)
        self.n_halo += n

    def _row_filters(self, table):
        """Property-level category filters, vectorised: the bound subhalo's particle counts per halo
        (subhalo_properties.py:2618-2626, category_filter.py:69-106)."""
        sub = self.props[0]
        cf = getattr(sub, "category_filter", None)
        if cf is None:
            return {}
        cols = layout.columns_of(sub.kind)
        blk = table[:, self.coloff[0] : self.coloff[1]]
        counts = {s: blk[:, cols[k][0]].astype(np.int64) for s, k in ((0, "Ngas"), (1, "Ndm"), (2, "Nstar"), (3, "Nbh"))}
        out = {}
        for fname in cf.filters:
            limit, mask = cf.engine_filter(fname)
            v = np.zeros(table.shape[0], dtype=np.int64)
            for s in range(4):
                if mask & (1 << s):
                    v += counts[s]
            out[fname] = v >= limit
        return out

    def _put(self, name, values, unit, desc, physical, a_exp):
        self._chunks.setdefault(name, []).append(values)
        self.meta.setdefault(name, {"unit": unit, "description": desc, "output_physical": bool(physical),
                                    "a_scale_exponent": a_exp, "dtype": str(values.dtype)})  # fmt: skip

    def append_packed(self, packed: "PackedLayout", narrow, wide, halo_arrays: Dict[str, np.ndarray]) -> None:
        """The same columns as `append_table`, from the packed catalogue (`PackedLayout.pack`, possibly gathered from
        several GPUs): host arrays narrow [n, Nn] float32, wide [n, Nw] float64."""
        narrow, wide = np.asarray(narrow), np.asarray(wide)
        n = narrow.shape[0]
        if n == 0:
            return
        for name, kind, c0, nc, dtype, unit, desc, physical, a_exp in packed.fields:
            if kind == "zero":
                vals = np.zeros((n, nc), dtype=dtype)
            elif kind == "narrow":
                vals = narrow[:, c0 : c0 + nc]
            else:
                vals = wide[:, c0 : c0 + nc].astype(dtype)
            self._put(name, vals[:, 0].copy() if nc == 1 else vals.copy(), unit, desc, physical, a_exp)
        for src, (name, dtype, desc, phys, aexp) in _INPUT_HALO_COLUMNS.items():
            unit = "snap_length" if src == "cofp" else "dimensionless"
            self._put(name, np.asarray(halo_arrays[src]).astype(dtype), unit, desc, phys, aexp)
        self.n_halo += n

    # -- reading ------------------------------------------------------------------------------
    def columns(self) -> Dict[str, np.ndarray]:
        return {k: np.concatenate(v, axis=0) for k, v in self._chunks.items()}

    def sorted_columns(self, cell_size, dimension) -> Dict[str, np.ndarray]:
        """Final catalogue order: SWIFT cell of the halo centre, then catalogue index
        (combine_chunks.py:38-52)."""
        from ..catalogue import spatial_sort_order

        cols = self.columns()
        order = spatial_sort_order(cols["InputHalos/HaloCentre"], cols["InputHalos/HaloCatalogueIndex"], cell_size, dimension)
        return {k: v[order] for k, v in cols.items()}

    def write(self, path: str, cell_size=None, dimension=None) -> None:
        """`path`.npz with one array per dataset and `path`.json with unit / description / physical /
        a-exponent per dataset (h5py is not available in this environment; the names and dtypes are the
        reference's, so a converter to the SOAP HDF5 layout is a loop over the keys)."""
        cols = self.sorted_columns(cell_size, dimension) if cell_size is not None else self.columns()
        np.savez(path + ".npz", **{k.replace("/", "|"): v for k, v in cols.items()})
        with open(path + ".json", "w") as f:
            json.dump(self.meta, f, indent=1, sort_keys=True)


class PackedLayout:
    """
    Which columns of the engine's fp64 table end up in the catalogue, and as what: what `ResultSet.append_table` does
    on the host, turned into index lists so that the cut, the comoving -> stored-unit factor and the cast can run on the
    device BEFORE the catalogue leaves the GPU (the all-gather then moves the stored catalogue -- float32 for most
    datasets, about a quarter of the bytes of the fp64 table for a dark-matter-only run -- instead of every column of the
    table).  `narrow`: datasets stored as float32; `wide`: float64 datasets and the integer ones (carried as float64,
    exact, cast when the columns are cut out).
    """

    def __init__(self, halo_prop_list: List, engine_desc_columns):
        props = list(halo_prop_list)
        ncols = [int(engine_desc_columns(hp.calc_desc(props, i))) for i, hp in enumerate(props)]
        coloff = np.concatenate([[0], np.cumsum(ncols)]).astype(np.int64)
        self.table_columns = int(coloff[-1])
        narrow_cols, narrow_scale, wide_cols, wide_scale = [], [], [], []
        self.fields = []   # (dataset name, kind, first column in its matrix, width, dtype, unit, description, physical, a-exponent)
        for ci, hp in enumerate(props):
            cols = layout.columns_of(hp.kind)
            for iproj, (proj, base) in enumerate(_projections(hp)):
                for key in hp.enabled_properties():
                    prop = full_property_list[key]
                    filt = hp.property_filters.get(prop.name, "basic")
                    if ci > 0 and filt != "basic":
                        raise NotImplementedError(f"property-level category filter {filt!r} of {hp.group_name}/{prop.name}: use ResultSet.append_table")
                    name = f"{hp.group_name}{proj}/{prop.name}"
                    meta = (prop.unit, prop.description, prop.output_physical, prop.a_scale_exponent)
                    if key in layout.VIRTUAL_ZERO_KEYS:
                        self.fields.append((name, "zero", 0, prop.shape, prop.dtype) + meta)
                        continue
                    if hp.kind == layout.CALC_PROJECTED:
                        c0, nc = layout.projected_columns(key, iproj)
                    else:
                        c0, nc = cols[key]
                        c0 += base
                    idx = [int(coloff[ci]) + c0 + k for k in range(nc)]
                    if np.dtype(prop.dtype) == np.float32:
                        self.fields.append((name, "narrow", len(narrow_cols), nc, prop.dtype) + meta)
                        narrow_cols += idx
                        narrow_scale += [hp.a**prop.a_power if prop.a_power else 1.0] * nc
                    else:
                        self.fields.append((name, "wide", len(wide_cols), nc, prop.dtype) + meta)
                        wide_cols += idx
                        integer = np.issubdtype(np.dtype(prop.dtype), np.integer)
                        wide_scale += [1.0 if (integer or not prop.a_power) else hp.a**prop.a_power] * nc
        self.narrow_cols = np.asarray(narrow_cols, dtype=np.int64)
        self.narrow_scale = np.asarray(narrow_scale, dtype=np.float64)
        self.wide_cols = np.asarray(wide_cols, dtype=np.int64)
        self.wide_scale = np.asarray(wide_scale, dtype=np.float64)
        self._dev = {}

    def bytes_per_halo(self) -> int:
        return 4 * len(self.narrow_cols) + 8 * len(self.wide_cols)

    def pack(self, table):
        """table [n, table_columns] float64 (torch tensor on any device, or ndarray) -> (narrow float32 [n, Nn], wide
        float64 [n, Nw]): value * a**a_power in float64, then ONE cast -- what `_cast` does on the host."""
        if isinstance(table, np.ndarray):
            narrow = np.zeros((table.shape[0], len(self.narrow_cols)), dtype=np.float32)
            narrow += table[:, self.narrow_cols] * self.narrow_scale[None, :]
            return narrow, table[:, self.wide_cols] * self.wide_scale[None, :]
        import torch

        key = str(table.device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.as_tensor(x, device=table.device) for x in
                                   (self.narrow_cols, self.narrow_scale, self.wide_cols, self.wide_scale))
        nc, ns, wc, ws = self._dev[key]
        narrow = (table.index_select(1, nc) * ns[None, :]).to(torch.float32)
        wide = table.index_select(1, wc) * ws[None, :]
        return narrow, wide


    def pack_engine(self, engine, device, order=None, n_rows=None):
        """`pack` of the table of the engine's last run without copying it out first: the engine writes the stored
        columns, scaled and cast, straight into the two matrices -- in the row order `order` (a torch int64 tensor on the
        device, e.g. the final-catalogue order; None = the table's own).  Same values as `pack(table)[order]` bit for bit."""
        import torch

        key = ("engine", str(device))
        if key not in self._dev:
            self._dev[key] = (torch.as_tensor(self.narrow_cols.astype(np.int32), device=device), torch.as_tensor(self.narrow_scale, device=device),
                              torch.as_tensor(self.wide_cols.astype(np.int32), device=device), torch.as_tensor(self.wide_scale, device=device))
        nc, ns, wc, ws = self._dev[key]
        n = int(order.numel()) if order is not None else int(n_rows if n_rows is not None else engine.n_halo)
        narrow = torch.empty((n, len(self.narrow_cols)), dtype=torch.float32, device=device)
        wide = torch.empty((n, len(self.wide_cols)), dtype=torch.float64, device=device)
        engine.pack_catalogue(order, nc, ns, wc, ws, narrow, wide)
        return narrow, wide


def _projections(hp):
    """('', 0) for ordinary calculations; ('/projx', 0), ('/projy', NC), ('/projz', 2 NC) for projected apertures
    (projected_aperture_properties.py:1891-1999: one group per projection axis)."""
    if hp.kind == layout.CALC_PROJECTED:
        return [(f"/{p}", i * layout.PROJECTED_NCOLS) for i, p in enumerate(("projx", "projy", "projz"))]
    return [("", 0)]


def _cast(raw, prop, a):
    """One cast to the table dtype, the way the reference fills a zero-initialised array of that dtype
    (SO_properties.py:3600-3620, :3682-3699)."""
    raw = np.asarray(raw, dtype=np.float64)
    if np.issubdtype(prop.dtype, np.integer):
        out = raw.astype(prop.dtype)
    else:
        out = np.zeros(raw.shape, dtype=prop.dtype)
        out += raw * (a**prop.a_power) if prop.a_power else raw
    return out[:, 0] if prop.shape == 1 else out


def run_chunk(engine, halo_prop_list: List, cellgrid, halo_arrays: Dict, max_passes: int = 8, allow_unfinished: bool = False) -> ResultSet:
    """
    Process every halo of a staged chunk: repeat `process_halos` while halos ask for a larger read radius
    (chunk_tasks.py:306-340: the reference re-reads a larger region and calls process_halos again; here
    the chunk's particles are already resident, so a pass just re-runs the unfinished halos with the radii
    the engine returned).  Returns the ResultSet of all finished halos.  A halo that is still not done at the end -- its read
    radius has outgrown what is staged, or `max_passes` is spent -- is an error (the reference loops until none is left; a
    catalogue with rows missing silently is worse than none) unless `allow_unfinished` says the caller looks at
    `halo_arrays["done"]` itself.
    """
    from . import halo_tasks

    rs = ResultSet(halo_prop_list, engine.desc_columns)
    halo_arrays.setdefault("done", np.zeros(len(halo_arrays["index"]), dtype=np.int32))
    for _ in range(max_passes):
        if not np.any(halo_arrays["done"] == 0):
            break
        before = int(np.sum(halo_arrays["done"]))
        out = halo_tasks.process_halos(engine, halo_prop_list, cellgrid, halo_arrays, None, unpack=False)
        raw = out[4]
        ok = raw["status"] == halo_tasks.STATUS_OK
        rows = raw["rows"][ok]
        if rows.size:
            table = engine.fetch(table=True)["table"][ok]
            rs.append_table(table, {k: np.asarray(halo_arrays[k])[rows] for k in _INPUT_HALO_COLUMNS})
        if int(np.sum(halo_arrays["done"])) == before and not np.any(raw["status"] == halo_tasks.STATUS_NEED_LARGER_READ_RADIUS):
            break
    nleft = int(np.sum(halo_arrays["done"] == 0))
    if nleft and not allow_unfinished:
        raise RuntimeError(f"{nleft} halos of the staged chunk are not finished after {max_passes} passes")
    return rs
