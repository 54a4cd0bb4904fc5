"""
Chunk-level runtime around the engine: the counterpart of `ChunkTask.__call__`
(SOAP/core/chunk_tasks.py:189-367), `mask_cells` (SOAP/core/mask_cells.py:6-38) and the
cell-masked read of `SWIFTCellGrid` (SOAP/core/swift_cells.py:540-549, :551-737), without HDF5.

A SWIFT snapshot stores every particle type sorted by top-level cell (swift_cells.py:480-534
plans reads as contiguous (file, range) pieces per cell).  `SnapshotCells` is that layout for
arrays that are already in memory (host NumPy or torch tensors on the GPU): per type the
particle arrays in top-level-cell order plus `cell_offset[ncell + 1]`.  A chunk then gets
exactly what the reference reads for it: every particle of every top-level cell that overlaps
`[centre - read_radius - cell/2, centre + read_radius + cell/2]` of one of its unfinished halos
(mask_cells.py:30-34), concatenated in cell order -- boundary cells are replicated between
chunks, there is no particle exchange during the computation.

`run_chunk_task` is the loop of chunk_tasks.py:189-367: mask cells -> read -> (box wrap) -> stage
(= mesh build) -> process_halos, repeated with the enlarged read radii until no halo is left.
torch is only plumbing here (index arithmetic and gathers of device-resident arrays).
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import halo_tasks
from .halo_tasks import _INPUT_HALO_COLUMNS
from .result_set import ResultSet


def box_wrap(pos, ref_pos, boxsize):
    """Periodic copies nearest `ref_pos` (chunk_tasks.py:48-50), NumPy or torch."""
    if hasattr(pos, "is_cuda"):
        import torch

        shift = torch.as_tensor(np.asarray(ref_pos, dtype=np.float64) - 0.5 * boxsize, device=pos.device)
        return (pos - shift[None, :]) % boxsize + shift[None, :]
    shift = np.asarray(ref_pos, dtype=np.float64)[None, :] - 0.5 * boxsize
    return (pos - shift) % boxsize + shift


def mask_cells(cell_size, dimension, centre, radius, done) -> np.ndarray:
    """
    Top-level cells to read for the halos with done == 0 (mask_cells.py:6-38 with
    swift_cells.py:540-549): for every such halo the index range
    floor((c - r - cell/2) / cell) .. floor((c + r + cell/2) / cell) per dimension, wrapped
    periodically.  Vectorised over halos; returns a bool array of shape `dimension`.
    """
    dimension = tuple(int(d) for d in dimension)
    cell_size = np.broadcast_to(np.asarray(cell_size, dtype=np.float64), (3,))
    centre = np.asarray(centre, dtype=np.float64).reshape(-1, 3)
    radius = np.asarray(radius, dtype=np.float64)
    todo = np.asarray(done) == 0
    mask = np.zeros(dimension, dtype=bool)
    if not np.any(todo):
        return mask
    c, r = centre[todo], radius[todo]
    pos_min = c - r[:, None] - 0.5 * cell_size[None, :]
    pos_max = c + r[:, None] + 0.5 * cell_size[None, :]
    imin = np.floor(pos_min / cell_size[None, :]).astype(np.int64)
    imax = np.floor(pos_max / cell_size[None, :]).astype(np.int64)
    # coverage count of the boxes by a 3-D difference array on the doubled grid (a box starts inside the first period and is at
    # most one period long), folded back onto the periodic grid: O(halos + cells) instead of a [halo, cell] table
    dims = np.array(dimension, dtype=np.int64)
    span = np.minimum(imax - imin + 1, dims[None, :])
    lo = imin % dims[None, :]
    hi = lo + span
    diff = np.zeros(tuple(2 * dims + 1), dtype=np.int32)
    for sx, x in ((1, lo[:, 0]), (-1, hi[:, 0])):
        for sy, y in ((1, lo[:, 1]), (-1, hi[:, 1])):
            for sz, z in ((1, lo[:, 2]), (-1, hi[:, 2])):
                np.add.at(diff, (x, y, z), sx * sy * sz)
    # (in place: three int64 temporaries of the doubled grid otherwise; a cell is covered by at most len(centre) boxes)
    if centre.shape[0] >= 2**31:
        diff = diff.astype(np.int64)
    for axis in range(3):
        np.cumsum(diff, axis=axis, out=diff)
    cnt = diff[:-1, :-1, :-1]
    d0, d1, d2 = dimension
    folded = cnt.reshape(2, d0, 2, d1, 2, d2).sum(axis=(0, 2, 4))
    return folded > 0


def resident_radius(mask: np.ndarray, cell_size, centre, read_radius, factor: float = 1.5, steps: int = 8) -> np.ndarray:
    """
    How far around each halo centre the cells of `mask` are complete: the largest of read_radius * factor^k
    (k = 0 .. steps) whose box -- with the half-cell margin of mask_cells.py:30-34 -- lies inside flagged cells.
    The engine may then apply read-radius updates (halo_tasks.py:393-402) that stay within it without a re-read
    (soap_engine_set_resident_radius).  A fully flagged periodic box is complete to any radius.
    """
    mask = np.asarray(mask, dtype=bool)
    centre = np.asarray(centre, dtype=np.float64).reshape(-1, 3)
    rr = np.asarray(read_radius, dtype=np.float64)
    if mask.all():
        return np.full(rr.shape, np.inf)
    dim = np.array(mask.shape)
    cs = np.broadcast_to(np.asarray(cell_size, dtype=np.float64), (3,))
    # number of cells NOT flagged in any box of the 3x tiled grid, from a summed-volume table
    sv_dtype = np.int32 if 27 * mask.size < 2**31 else np.int64   # (the table's last entry counts up to 27 x ncell cells)
    miss = np.tile(~mask, (3, 3, 3)).astype(sv_dtype)
    sv = np.zeros(tuple(3 * dim + 1), dtype=sv_dtype)
    np.cumsum(miss, axis=0, out=miss)
    np.cumsum(miss, axis=1, out=miss)
    np.cumsum(miss, axis=2, out=sv[1:, 1:, 1:])

    def missing(lo, hi):   # inclusive index boxes in the tiled grid
        x0, y0, z0 = lo[:, 0], lo[:, 1], lo[:, 2]
        x1, y1, z1 = hi[:, 0] + 1, hi[:, 1] + 1, hi[:, 2] + 1
        return (sv[x1, y1, z1] - sv[x0, y1, z1] - sv[x1, y0, z1] - sv[x1, y1, z0]
                + sv[x0, y0, z1] + sv[x0, y1, z0] + sv[x1, y0, z0] - sv[x0, y0, z0])  # fmt: skip

    out = rr.copy()
    alive = np.ones(rr.shape, dtype=bool)
    r = rr.copy()
    for k in range(steps + 1):
        imin = np.floor((centre - r[:, None] - 0.5 * cs[None, :]) / cs[None, :]).astype(np.int64)
        imax = np.floor((centre + r[:, None] + 0.5 * cs[None, :]) / cs[None, :]).astype(np.int64)
        fits = np.all(imax - imin + 1 <= dim[None, :], axis=1) & np.all(imin >= -dim[None, :], axis=1) & np.all(imax < 2 * dim[None, :], axis=1)
        lo = np.where(fits[:, None], imin + dim[None, :], 0)
        hi = np.where(fits[:, None], imax + dim[None, :], 0)
        covered = fits & (missing(lo, hi) == 0)
        alive &= covered
        out = np.where(alive, r, out)
        r = r * factor
    return out


class SnapshotCells:
    """
    Particle arrays in top-level-cell order, the in-memory stand-in of a SWIFT snapshot
    (swift_cells.py:149-404 keeps `cell_size`, `dimension` and per-type cell offsets/counts).
    data[ptype] = {"Coordinates", "Velocities", "Masses", "GroupNr_bound", "FOFGroupIDs", ...};
    arrays are NumPy or torch (any device).  Particles must already be sorted by
    cell index ix * dim1 * dim2 + iy * dim2 + iz of floor(pos / cell_size); `from_unsorted` sorts.
    """

    def __init__(self, data: Dict, boxsize: float, dimension: Sequence[int]):
        self.data = data
        self.boxsize = float(boxsize)
        self.dimension = tuple(int(d) for d in dimension)
        self.cell_size = np.array([self.boxsize / d for d in self.dimension])
        self.ncell = int(np.prod(self.dimension))
        self.cell_offset = {}
        for pt, d in data.items():
            cid = self._cell_index(d["Coordinates"])
            if _is_torch(cid):
                import torch

                if cid.numel() > 1 and bool((cid[1:] < cid[:-1]).any()):
                    raise ValueError("particles are not stored in top-level-cell order")
                counts = torch.bincount(cid, minlength=self.ncell)
                off = torch.zeros(self.ncell + 1, dtype=torch.int64, device=cid.device)
                off[1:] = torch.cumsum(counts, 0)
            else:
                if cid.size > 1 and np.any(cid[1:] < cid[:-1]):
                    raise ValueError("particles are not stored in top-level-cell order")
                off = np.zeros(self.ncell + 1, dtype=np.int64)
                off[1:] = np.cumsum(np.bincount(cid, minlength=self.ncell))
            self.cell_offset[pt] = off

    def _cell_index(self, pos):
        d0, d1, d2 = self.dimension
        if _is_torch(pos):
            import torch

            cs = torch.as_tensor(self.cell_size, device=pos.device)
            dim = torch.as_tensor(self.dimension, device=pos.device)
            ic = torch.minimum(torch.clamp(torch.floor(pos / cs[None, :]).to(torch.int64), min=0), dim[None, :] - 1)
            return (ic[:, 0] * d1 + ic[:, 1]) * d2 + ic[:, 2]
        ic = np.clip(np.floor(pos / self.cell_size[None, :]).astype(np.int64), 0, np.array(self.dimension) - 1)
        return (ic[:, 0] * d1 + ic[:, 1]) * d2 + ic[:, 2]

    @classmethod
    def from_unsorted(cls, data: Dict, boxsize: float, dimension: Sequence[int]):
        tmp = cls.__new__(cls)
        tmp.boxsize = float(boxsize)
        tmp.dimension = tuple(int(d) for d in dimension)
        tmp.cell_size = np.array([tmp.boxsize / d for d in tmp.dimension])
        out = {}
        for pt, d in data.items():
            cid = tmp._cell_index(d["Coordinates"])
            if _is_torch(cid):
                import torch

                order = torch.argsort(cid, stable=True)
            else:
                order = np.argsort(cid, kind="stable")
            out[pt] = {k: v[order] for k, v in d.items()}
        return cls(out, boxsize, dimension)

    def nr_particles(self) -> int:
        return sum(int(d["Coordinates"].shape[0]) for d in self.data.values())

    def read_masked_cells(self, mask: np.ndarray) -> Dict:
        """Particles of the flagged cells, concatenated in cell order, for every type
        (read_masked_cells_to_shared_memory, swift_cells.py:551-737, minus the file I/O)."""
        flat = np.asarray(mask, dtype=bool).reshape(-1)
        out = {}
        for pt, d in self.data.items():
            off = self.cell_offset[pt]
            if bool(flat.all()):
                out[pt] = d   # the whole snapshot: nothing is copied
                continue
            if _is_torch(off):
                import torch

                m = torch.as_tensor(flat, device=off.device)
                starts = off[:-1][m]
                lengths = (off[1:] - off[:-1])[m]
                total = int(lengths.sum().item())
                dst0 = torch.cumsum(lengths, 0) - lengths
                idx = torch.repeat_interleave(starts - dst0, lengths, output_size=total) + torch.arange(total, device=off.device)
                out[pt] = {k: v.index_select(0, idx) for k, v in d.items()}
            else:
                starts = off[:-1][flat]
                lengths = (off[1:] - off[:-1])[flat]
                total = int(lengths.sum())
                dst0 = np.cumsum(lengths) - lengths
                idx = np.repeat(starts - dst0, lengths) + np.arange(total)
                out[pt] = {k: np.asarray(v)[idx] for k, v in d.items()}
        return out


def _is_torch(x):
    return hasattr(x, "is_cuda")


def chunk_ref_pos(centre) -> np.ndarray:
    """Reference position of the periodic wrap: midpoint of the bounding box of the chunk's halo centres
    (chunk_tasks.py:213-217)."""
    centre = np.asarray(centre, dtype=np.float64).reshape(-1, 3)
    return (np.amin(centre, axis=0) + np.amax(centre, axis=0)) / 2


def read_chunk(snapshot: SnapshotCells, halo_arrays: Dict, wrap: bool = True, return_mask: bool = False):
    """Mask + read (+ box wrap) for the unfinished halos of `halo_arrays`.  Returns (data, nr_cells[, mask])."""
    done = halo_arrays.get("done")
    if done is None:
        done = np.zeros(len(halo_arrays["index"]), dtype=np.int32)
    mask = mask_cells(snapshot.cell_size, snapshot.dimension, halo_arrays["cofp"], halo_arrays["read_radius"], done)
    data = snapshot.read_masked_cells(mask)
    if wrap and not bool(mask.all()):
        ref = chunk_ref_pos(np.asarray(halo_arrays["cofp"]))
        data = {pt: dict(d, Coordinates=box_wrap(d["Coordinates"], ref, snapshot.boxsize)) for pt, d in data.items()}
    if return_mask:
        return data, int(mask.sum()), mask
    return data, int(mask.sum())


def run_chunk_task(engine, snapshot: SnapshotCells, halo_prop_list: List, cellgrid, halo_arrays: Dict,
                   wrap: bool = True, max_iterations: int = 12, results: Optional[ResultSet] = None,
                   log: Optional[List] = None, data: Optional[Dict] = None,
                   raw_tables: Optional[List] = None, use_resident_radius: bool = False) -> ResultSet:  # fmt: skip
    """
    Process every halo of one chunk (chunk_tasks.py:189-367).  `halo_arrays` holds the chunk's halo rows
    (index, cofp, search_radius, read_radius, is_central, nr_bound_part), already in processing order
    (most bound particles first, chunk_tasks.py:118-120); `done` is added.  Every iteration reads the
    cells the unfinished halos need with their current read radius, stages them (one mesh per particle
    type) and runs `process_halos`; halos that come back with a larger read radius (halo_tasks.py:393-402)
    go round again.  `data` short-cuts the first read (the region is already resident).
    `use_resident_radius=True` tells the engine how far the cells that were read are complete around every halo
    (`resident_radius`), so that read-radius updates inside the region need no further iteration.
    `raw_tables` collects, per iteration, the engine's fp64 rows of the finished halos with their
    catalogue columns and statistics (what the catalogue gather of SURVEY.md 8(e) concatenates).
    `wrap=False` leaves the coordinates as stored: the engine enumerates the periodic copies of a search
    sphere itself (shared_mesh.py:146-161), the wrap only keeps the mesh volume small.
    """
    rs = results if results is not None else ResultSet(halo_prop_list, engine.desc_columns)
    n = len(halo_arrays["index"])
    halo_arrays.setdefault("done", np.zeros(n, dtype=np.int32))
    for it in range(max_iterations):
        if not np.any(halo_arrays["done"] == 0):
            break
        resident = None
        if data is None:
            data, nr_cells, mask = read_chunk(snapshot, halo_arrays, wrap, return_mask=True)
            if use_resident_radius:
                resident = resident_radius(mask, snapshot.cell_size, halo_arrays["cofp"], halo_arrays["read_radius"])
        else:
            nr_cells = -1
        nr_parts = sum(int(d["Coordinates"].shape[0]) for d in data.values())
        if nr_parts == 0:
            raise RuntimeError("Task has zero particles?!")   # chunk_tasks.py:262-264
        halo_tasks.stage_chunk(engine, data)
        out = halo_tasks.process_halos(engine, halo_prop_list, cellgrid, halo_arrays, None, unpack=False,
                                       resident_radius=resident)
        raw = out[4]
        ok = raw["status"] == halo_tasks.STATUS_OK
        rows = raw["rows"][ok]
        if rows.size:
            table = engine.fetch(table=True)["table"][ok]
            rs.append_table(table, {k: np.asarray(halo_arrays[k])[rows] for k in _INPUT_HALO_COLUMNS})
            if raw_tables is not None:
                raw_tables.append(dict(table=table, stats=raw["stats"][ok], search_radius=raw["search_radius"][ok],
                                       **{k: np.asarray(halo_arrays[k])[rows] for k in _INPUT_HALO_COLUMNS}))
        if log is not None:
            log.append(dict(iteration=it, cells=nr_cells, particles=nr_parts, done=int(ok.sum()), left=out[2]))
        data = None   # a further iteration reads the (larger) region of the halos that are left
    nleft = int(np.sum(halo_arrays["done"] == 0))
    if nleft:
        # the reference loops until no halo is left (chunk_tasks.py:189-367); a catalogue that silently misses rows is worse
        # than no catalogue
        raise RuntimeError(f"{nleft} halos still ask for a larger read radius after {max_iterations} iterations "
                           f"(first catalogue index: {int(np.asarray(halo_arrays['index'])[halo_arrays['done'] == 0][0])})")
    return rs


def split_halos(halos: Dict[str, np.ndarray], boxsize: float, nr_chunks: int, separate_chunks=None, weights=None):
    """Halo tables of the chunks: Peano-Hilbert decomposition (domain_decomposition.py), then every chunk's
    halos in decreasing order of bound-particle count (chunk_tasks.py:118-120).  `weights`: equal total weight per chunk
    instead of the reference's equal halo count (`domain_decomposition.halo_work_weights`; "work" uses it)."""
    from .domain_decomposition import halo_work_weights, peano_decomposition

    if isinstance(weights, str) and weights == "work":
        weights = halo_work_weights(halos)
    ordered, chunk_size = peano_decomposition(boxsize, halos, nr_chunks, separate_chunks, weights=weights)
    out = []
    first = 0
    for size in chunk_size:
        part = {k: v[first : first + size] for k, v in ordered.items()}
        order = np.argsort(part["nr_bound_part"])[::-1]   # chunk_tasks.py:118
        out.append({k: v[order].copy() for k, v in part.items()})
        first += size
    return out
