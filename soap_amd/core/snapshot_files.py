"""
An HDF5-free on-disk container for what the hot path reads, and its cell-masked reader.

The reference reads particles with h5py from a SWIFT snapshot and its membership files
(SOAP/core/swift_cells.py:149-404 for the metadata, :480-534 for the read plan, :551-737 for the read) and the halo
table from an HBT-HERONS catalogue (SOAP/catalogue_readers/read_hbtplus.py:184-403).  h5py is not part of this engine:
a snapshot is exported ONCE to the layout below (INTEGRATION.md shows the h5py script that does it) and the chunk loop
reads from that.

    <dir>/index.json                     format tag, box size, top-level cell grid, header scalars, datasets per type
    <dir>/PartType<t>.<Dataset>.bin      raw little-endian C-order array, particles in top-level-cell order
    <dir>/PartType<t>.cell_offset.bin    int64[ncell + 1]: first particle of every top-level cell (row-major ix, iy, iz)
    <dir>/catalogue.json + catalogue.<column>.bin    the halo table

`SnapshotFiles` is a `SnapshotCells` whose arrays are `np.memmap`s: `read_masked_cells(mask)` plans the reads as the
reference does -- flagged non-empty cells in file order, adjacent cells merged into one piece (swift_cells.py:498-519) --
and copies the pieces into ONE host buffer per dataset (page-locked when a GPU is present, so that the engine's
`stage_particles` copy runs at PCIe rate), returned as NumPy views.  Nothing here computes; it is the I/O edge of
SURVEY.md section 8, row f4.
"""

from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, Optional, Sequence

import numpy as np

from ..units import SnapshotScalars
from .chunk_tasks import SnapshotCells

FORMAT = "soap_amd.snapshot/1"
CATALOGUE_FORMAT = "soap_amd.catalogue/1"
# the datasets the engine stages per particle type (chunk_tasks.SnapshotCells; black holes: DynamicalMasses as Masses)
REQUIRED = ("Coordinates", "Velocities", "Masses", "GroupNr_bound", "FOFGroupIDs")


def _le(dtype) -> str:
    """little-endian dtype string of the file ('<f8', '<i4', ...)"""
    dt = np.dtype(dtype)
    return dt.newbyteorder("<").str if dt.itemsize > 1 else dt.str


def _host(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "is_cuda") else np.asarray(x)


def write_snapshot(path: str, data: Dict, scalars: SnapshotScalars, dimension: Sequence[int], extra_header: Optional[Dict] = None) -> None:
    """
    Export particle arrays to the container.  data[ptype] = {"Coordinates", "Velocities", "Masses", "GroupNr_bound",
    "FOFGroupIDs", ...} (NumPy or torch, any order): every type is sorted by top-level cell (stable, so particles of a cell
    keep their snapshot order, as in a SWIFT snapshot) and written dataset by dataset.
    """
    os.makedirs(path, exist_ok=True)
    cells = SnapshotCells.from_unsorted({pt: {k: _host(v) for k, v in d.items()} for pt, d in data.items()}, scalars.boxsize, dimension)
    index = {
        "format": FORMAT, "boxsize": float(scalars.boxsize), "dimension": [int(d) for d in cells.dimension],
        "header": dict(dataclasses.asdict(scalars), **(extra_header or {})), "types": {},
    }  # fmt: skip
    for pt, d in cells.data.items():
        t = int(str(pt)[-1]) if not isinstance(pt, int) else int(pt)
        missing = [k for k in REQUIRED if k not in d and not (k == "Masses" and "DynamicalMasses" in d)]
        if missing:
            raise ValueError(f"PartType{t}: datasets missing: {missing}")
        entry = {"n": int(d["Coordinates"].shape[0]), "datasets": {}, "cell_offset": f"PartType{t}.cell_offset.bin"}
        np.asarray(cells.cell_offset[pt], dtype="<i8").tofile(os.path.join(path, entry["cell_offset"]))
        for name, arr in d.items():
            arr = np.ascontiguousarray(arr)
            fn = f"PartType{t}.{name}.bin"
            arr.astype(_le(arr.dtype), copy=False).tofile(os.path.join(path, fn))
            entry["datasets"][name] = {"file": fn, "dtype": _le(arr.dtype), "shape": [int(s) for s in arr.shape]}
        index["types"][str(t)] = entry
    tmp = os.path.join(path, "index.json.tmp")
    with open(tmp, "w") as f:
        json.dump(index, f, indent=1)
    os.replace(tmp, os.path.join(path, "index.json"))   # the index appears last: a directory with an index is complete


def write_catalogue(path: str, index, cofp, r_enclose_comoving, is_central, nr_bound_part, extra: Optional[Dict] = None) -> None:
    """
    The halo table the chunk loop needs (read_hbtplus.py:300-370: HaloCatalogueIndex, centre of potential = comoving most
    bound position, REncloseComoving, Rank == 0 as the central flag, Nbound), in snapshot length units, one raw file per column.
    """
    os.makedirs(path, exist_ok=True)
    cols = {"index": np.asarray(index, dtype="<i8"), "cofp": np.asarray(cofp, dtype="<f8").reshape(-1, 3),
            "r_enclose_comoving": np.asarray(r_enclose_comoving, dtype="<f8"), "is_central": np.asarray(is_central, dtype="<i4"),
            "nr_bound_part": np.asarray(nr_bound_part, dtype="<i8")}  # fmt: skip
    for k, v in (extra or {}).items():
        cols[k] = np.ascontiguousarray(_host(v))
    n = len(cols["index"])
    meta = {"format": CATALOGUE_FORMAT, "n": int(n), "columns": {}}
    for k, v in cols.items():
        if v.shape[0] != n:
            raise ValueError(f"catalogue column {k} has {v.shape[0]} rows, the table {n}")
        fn = f"catalogue.{k}.bin"
        v.astype(_le(v.dtype), copy=False).tofile(os.path.join(path, fn))
        meta["columns"][k] = {"file": fn, "dtype": _le(v.dtype), "shape": [int(s) for s in v.shape]}
    tmp = os.path.join(path, "catalogue.json.tmp")
    with open(tmp, "w") as f:
        json.dump(meta, f, indent=1)
    os.replace(tmp, os.path.join(path, "catalogue.json"))


def read_catalogue(path: str, min_read_radius: float = 0.0) -> Dict[str, np.ndarray]:
    """
    `halo_arrays` of the chunk loop from a catalogue table: search_radius = 1.01 x REncloseComoving
    (read_hbtplus.py:319-321), read_radius = the same, not below `min_read_radius` (halo_centres.py:188-190).
    Further columns of the table (e.g. TrackId) come back under their own names.
    """
    with open(os.path.join(path, "catalogue.json")) as f:
        meta = json.load(f)
    if meta.get("format") != CATALOGUE_FORMAT:
        raise ValueError(f"{path}: not a {CATALOGUE_FORMAT} table")
    cols = {}
    for k, c in meta["columns"].items():
        a = np.fromfile(os.path.join(path, c["file"]), dtype=c["dtype"])
        if a.size != int(np.prod(c["shape"])):
            raise ValueError(f"{c['file']}: {a.size} values on disk, {c['shape']} in the index")
        cols[k] = a.reshape(c["shape"]).astype(np.dtype(c["dtype"]).newbyteorder("="), copy=False)
    search = 1.01 * cols.pop("r_enclose_comoving")
    halos = dict(cols, search_radius=search, read_radius=np.maximum(search, float(min_read_radius)))
    halos["is_central"] = halos["is_central"].astype(np.int32)
    return halos


class _HostBuffers:
    """One growing destination buffer per (type, dataset); page-locked through torch when a GPU is there (torch is only the
    allocator: the reader hands out NumPy views)."""

    def __init__(self, pinned: Optional[bool]):
        self._buf = {}
        if pinned is None:
            try:
                import torch

                pinned = bool(torch.cuda.is_available())
            except Exception:
                pinned = False
        self.pinned = pinned

    def get(self, key, n: int, tail: tuple, dtype) -> np.ndarray:
        dt = np.dtype(dtype)
        need = int(n) * int(np.prod(tail, dtype=np.int64)) * dt.itemsize
        cur = self._buf.get(key)
        if cur is None or cur.nbytes < need:
            cap = max(need + need // 4, 1)   # headroom: the next iteration reads a somewhat larger region
            if self.pinned:
                import torch

                self._keep = getattr(self, "_keep", {})
                t = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
                self._keep[key] = t
                cur = t.numpy()
            else:
                cur = np.empty(cap, dtype=np.uint8)
            self._buf[key] = cur
        return cur[:need].view(dt).reshape((int(n),) + tuple(tail))


class SnapshotFiles(SnapshotCells):
    """
    The container opened for reading: per type the datasets as `np.memmap` (nothing is read until a chunk asks for cells)
    and `cell_offset`.  Drop-in for `SnapshotCells` in `read_chunk` / `run_chunk_task`.  The arrays `read_masked_cells`
    returns are views of this object's host buffers: they are valid until the next `read_masked_cells` call, which is all
    the chunk loop needs (`stage_particles` has copied them by the time it returns, include/soap_engine.h).
    """

    def __init__(self, path: str, pinned: Optional[bool] = None, datasets: Optional[Sequence[str]] = None):
        with open(os.path.join(path, "index.json")) as f:
            index = json.load(f)
        if index.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} container")
        self.path = path
        self.boxsize = float(index["boxsize"])
        self.dimension = tuple(int(d) for d in index["dimension"])
        self.cell_size = np.array([self.boxsize / d for d in self.dimension])
        self.ncell = int(np.prod(self.dimension))
        self.header = index["header"]
        self.data, self.cell_offset = {}, {}
        for t, entry in index["types"].items():
            pt = int(t)
            off = np.fromfile(os.path.join(path, entry["cell_offset"]), dtype="<i8").astype(np.int64)
            if off.size != self.ncell + 1 or off[0] != 0 or off[-1] != entry["n"] or np.any(np.diff(off) < 0):
                raise ValueError(f"PartType{t}: cell offsets do not describe {entry['n']} particles in {self.ncell} cells")
            self.cell_offset[pt] = off
            d = {}
            for name, c in entry["datasets"].items():
                if datasets is not None and name not in datasets:
                    continue
                shape = tuple(c["shape"])
                if shape[0] != entry["n"]:
                    raise ValueError(f"{c['file']}: {shape[0]} rows, the type has {entry['n']} particles")
                fn = os.path.join(path, c["file"])
                if os.path.getsize(fn) != int(np.prod(shape, dtype=np.int64)) * np.dtype(c["dtype"]).itemsize:
                    raise ValueError(f"{c['file']}: size on disk does not match {shape} of {c['dtype']}")
                d[name] = np.memmap(fn, dtype=c["dtype"], mode="r", shape=shape) if shape[0] > 0 else np.empty(shape, dtype=c["dtype"])
            self.data[pt] = d
        self._buffers = _HostBuffers(pinned)
        self.bytes_read = 0
        self.pieces_read = 0

    def scalars(self) -> SnapshotScalars:
        """the header scalars as the engine's SnapshotScalars (swift_cells.py:224-287 keeps the same on the cell grid)"""
        names = {f.name for f in dataclasses.fields(SnapshotScalars)}
        return SnapshotScalars(**{k: v for k, v in self.header.items() if k in names})

    def plan_read(self, pt, mask: np.ndarray):
        """(first particle, count) pieces of the flagged cells of one type, in file order, adjacent cells merged
        (prepare_read, swift_cells.py:480-534; there is one file per dataset here, so a piece is never split)."""
        flat = np.asarray(mask, dtype=bool).reshape(-1)
        off = self.cell_offset[pt]
        starts, counts = off[:-1][flat], (off[1:] - off[:-1])[flat]
        keep = counts > 0
        starts, counts = starts[keep], counts[keep]
        if starts.size == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        new_piece = np.ones(starts.size, dtype=bool)
        new_piece[1:] = starts[1:] != starts[:-1] + counts[:-1]
        first = np.flatnonzero(new_piece)
        ends = np.append(first[1:], starts.size) - 1
        return starts[first], (starts[ends] + counts[ends]) - starts[first]

    def read_masked_cells(self, mask: np.ndarray) -> Dict:
        out = {}
        for pt, d in self.data.items():
            p0, pn = self.plan_read(pt, mask)
            total = int(pn.sum())
            dst0 = np.cumsum(pn) - pn
            arrays = {}
            for name, src in d.items():
                native = np.dtype(src.dtype).newbyteorder("=")
                dst = self._buffers.get((pt, name), total, src.shape[1:], native)
                for s, n, o in zip(p0, pn, dst0):
                    dst[o : o + n] = src[s : s + n]
                arrays[name] = dst
                self.bytes_read += dst.nbytes
            self.pieces_read += int(p0.size) * len(d)
            out[pt] = arrays
        return out
