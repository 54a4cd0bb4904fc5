"""
Chunk-level scheduler: the drop-in for `process_halos` of SOAP/core/halo_tasks.py:276-430.

The reference pulls one halo at a time off an MPI RMA counter and runs
process_single_halo() (halo_tasks.py:23-273) in Python; here the whole halo table of the
chunk goes to the HIP engine in one call (the search-radius loop, the calculation list
and the retry rules all run on the device, one workgroup per halo) and the result table
is unpacked into the same `halo_result` dictionaries.
"""

from __future__ import annotations

import time
from typing import Dict, List

import numpy as np

from ..engine import (STATUS_NEED_LARGER_READ_RADIUS, STATUS_OK, Engine, SLOT_OF_PTYPE,
                      mesh_resolution)  # fmt: skip
from ..particle_selection.halo_properties import SearchRadiusTooSmallError  # noqa: F401

# Factors of the reference (halo_tasks.py:14-20); applied on the device, quoted here for
# the host-side bookkeeping and the documentation.
SEARCH_RADIUS_FACTOR = 1.2
READ_RADIUS_FACTOR = 1.5
REPORT_RADIUS = 20.0

_FATAL_TEXT = {
    -1: "Found more particles than expected for halo {index}",  # subhalo_properties.py:2642-2646
    -2: "Cannot find SO radius, but search radius is already larger than 20 Mpc! (halo {index})",
    -3: "search sphere of halo {index} does not fit the engine's slab",
    -4: "current_radius > read_radius for halo {index} (halo_tasks.py:78)",
}

_INPUT_HALO_COLUMNS = {
    # halo_arrays name -> (dataset name, dtype, description, physical, a-exponent)
    "cofp": ("InputHalos/HaloCentre", np.float64,
             "Centre of the subhalo as given by the halo finder.", False, 1),
    "index": ("InputHalos/HaloCatalogueIndex", np.int64,
              "Index of this halo in the halo finder catalogue.", True, None),
    "is_central": ("InputHalos/IsCentral", np.int64,
                   "Whether the halo finder flagged the halo as central (1) or satellite (0).", True, None),
    "nr_bound_part": ("InputHalos/NumberOfBoundParticles", np.int64,
                      "Total number of particles bound to the subhalo.", True, None),
}  # fmt: skip


# halo_tasks.py:203-227: present when the caller's halo_arrays carry them (float32 in the reference)
_TIMING_COLUMNS = {
    "n_loop": "Number of loops before target density was reached",
    "n_process": "Number of times this halo was processed (a halo will have to be reprocessed if it's target density "
                 "is not reached with the region currently loaded in memory)",
}  # fmt: skip


def target_density_of(halo_prop_list, cellgrid):
    """Minimum density any calculation needs (halo_tasks.py:306-317), converted once to
    mass / comoving length^3.  None if no calculation sets a density multiple."""
    target = None
    for halo_prop in halo_prop_list:
        if halo_prop.mean_density_multiple is not None:
            density = halo_prop.mean_density_multiple * cellgrid.mean_density
            if target is None or density < target:
                target = density
        if halo_prop.critical_density_multiple is not None:
            density = halo_prop.critical_density_multiple * cellgrid.critical_density
            if target is None or density < target:
                target = density
    return None if target is None else cellgrid.density_to_comoving(target)


def make_engine(cellgrid, device=0, **kw) -> Engine:
    """Engine configured from the cell grid scalars (what HaloProperty.__init__ copies)."""
    soft = cellgrid.softening_comoving()
    return Engine(
        boxsize=cellgrid.boxsize,
        G_eff=cellgrid.G_eff,
        softening=[soft] * 4,
        nu_density=cellgrid.nu_density_comoving,
        mpc20=cellgrid.mpc_to_comoving(20.0),
        a=cellgrid.a,
        mpc=1000.0 * cellgrid.kpc,
        G_phys=cellgrid.G,
        hubble=cellgrid.H,
        rdot_omega=2 * cellgrid.Omega_g + (3 / 2) * cellgrid.Omega_m,
        device=device,
        **kw,
    )


def stage_chunk(engine: Engine, data: Dict, resolution=None):
    """
    Put the chunk's particle arrays in HBM and build one cell-list mesh per particle type:
    the counterpart of share_array + SharedMesh(comm, pos, resolution) in
    ChunkTask.__call__ (SOAP/core/chunk_tasks.py:165-170, :293-304).
    data[ptype] = {"Coordinates","Velocities","Masses"|"DynamicalMasses","GroupNr_bound",
    "FOFGroupIDs"}; ptype is 0/1/4/5 or "PartType0"...; data[6] = {"Coordinates","Masses","Weights"};
    arrays are NumPy or torch-CUDA.  Types absent from `data` are removed from the engine.
    """
    present = {int(str(p)[-1]) if not isinstance(p, int) else p for p in data}
    for pt in (0, 1, 4, 5, 6):
        if pt not in present:   # a type the previous chunk had and this one has not
            engine.clear_type("nu" if pt == 6 else SLOT_OF_PTYPE[pt])
    for ptype, d in data.items():
        pt = int(str(ptype)[-1]) if not isinstance(ptype, int) else ptype
        if pt == 6:   # neutrinos: coordinates, masses, weights -- SO mass profile only
            res = mesh_resolution(int(d["Coordinates"].shape[0])) if resolution is None else resolution
            engine.stage_neutrinos(d["Coordinates"], d["Masses"], d["Weights"], res)
            continue
        if pt not in SLOT_OF_PTYPE:
            continue
        n = int(d["Coordinates"].shape[0])
        mass = d["DynamicalMasses"] if (pt == 5 and "DynamicalMasses" in d) else d["Masses"]
        res = mesh_resolution(n) if resolution is None else resolution
        engine.stage_particles(SLOT_OF_PTYPE[pt], d["Coordinates"], d["Velocities"], mass,
                               d["GroupNr_bound"], d["FOFGroupIDs"], res)  # fmt: skip


class BatchResult:
    """One engine run over a halo table, addressable by catalogue index: what the per-halo
    `HaloProperty.calculate` facade reads (operator protocol)."""

    STATUS_NEED_LARGER_READ_RADIUS = STATUS_NEED_LARGER_READ_RADIUS

    def __init__(self, table, status, index, coloff, subhalo):
        self.table, self.status, self.coloff, self.subhalo = table, status, coloff, subhalo
        self._row = {int(ix): r for r, ix in enumerate(index)}

    def row_of(self, index: int) -> int:
        return self._row[index]


def bind_results(engine: Engine, halo_prop_list: List, raw: Dict, halo_arrays: Dict) -> BatchResult:
    """Make the result of `process_halos(..., unpack=...)` available to `prop.calculate(input_halo, ...)`
    for every entry of halo_prop_list.  `raw` is the last element process_halos returned."""
    calcs = [hp.calc_desc(halo_prop_list, i) for i, hp in enumerate(halo_prop_list)]
    coloff = np.concatenate([[0], np.cumsum([engine.desc_columns(c) for c in calcs])])
    table = raw["table"] if raw.get("table") is not None else engine.fetch(table=True)["table"]
    index = np.asarray(halo_arrays["index"])[raw["rows"]]
    batch = BatchResult(table, raw["status"], index, coloff, halo_prop_list[0])
    for i, hp in enumerate(halo_prop_list):
        hp.bind(batch, i)
    return batch


def unpack_row(halo_prop_list: List, coloff, row) -> Dict:
    """One row of the engine's result table as the reference's `halo_result` (every calculation's `unpack` in list order;
    `coloff`: first column of each calculation's block)."""
    halo_result = {}
    do_calc = None
    for ci, hp in enumerate(halo_prop_list):
        hp.unpack(row[coloff[ci] : coloff[ci + 1]], halo_result, do_calc)
        if ci == 0 and getattr(hp, "category_filter", None) is not None:
            # property-level filters use the bound subhalo's particle counts
            # (subhalo_properties.py:2618-2626)
            do_calc = hp.category_filter.get_do_calculation(halo_result)
    return halo_result


def process_halos(engine: Engine, halo_prop_list: List, cellgrid, halo_arrays: Dict, results,
                  unpack: bool = True, resident_radius=None):  # fmt: skip
    """
    Compute every calculation of `halo_prop_list` for every halo of `halo_arrays` with
    done == 0.  halo_arrays: dict of NumPy arrays index, cofp, search_radius, read_radius,
    is_central, nr_bound_part, done (already in processing order).  Finished halos get
    done=1 and one `halo_result` dict appended to `results` (anything with .append);
    halos that need a larger region get their read_radius/search_radius updated exactly as
    the reference does (halo_tasks.py:393-402) and stay at done=0.

    `resident_radius` (scalar or per halo of halo_arrays; not in the reference): radius about each centre inside
    which the staged particles are known to be complete.  Read-radius updates that stay inside it are applied by
    the engine itself, which then starts the halo again as the next call would (halo_tasks.py:393-402 +
    chunk_tasks.py:189-367) -- the halo comes back finished, with the updated radii, instead of not done.

    Returns (total_time, task_time, nr_halos_left, nr_done, raw) where raw holds the
    engine's table/status/stats for callers that do not want per-halo dictionaries.
    """
    t0_all = time.time()
    todo = np.flatnonzero(np.asarray(halo_arrays["done"]) == 0)
    calcs = [hp.calc_desc(halo_prop_list, i) for i, hp in enumerate(halo_prop_list)]
    target_density = target_density_of(halo_prop_list, cellgrid)
    # first pass over a chunk: every halo is still to do, the catalogue arrays go in as they are
    pick = (lambda a: np.asarray(a)) if todo.size == len(halo_arrays["done"]) else (lambda a: np.asarray(a)[todo])
    if resident_radius is not None:
        resident_radius = np.broadcast_to(np.asarray(resident_radius, dtype=np.float64), (len(halo_arrays["done"]),))
        resident_radius = np.maximum(pick(resident_radius), pick(halo_arrays["read_radius"]))
    engine.stage_halos(
        pick(halo_arrays["cofp"]), pick(halo_arrays["search_radius"]),
        pick(halo_arrays["read_radius"]), pick(halo_arrays["index"]),
        pick(halo_arrays["is_central"]), pick(halo_arrays["nr_bound_part"]),
        resident_radius=resident_radius,
    )  # fmt: skip
    t0_task = time.time()
    engine.run(calcs, target_density)
    out = engine.fetch(table=unpack)
    task_time = time.time() - t0_task

    status = out["status"]
    fatal = np.flatnonzero(status < 0)
    if fatal.size:
        i = int(fatal[0])
        index = int(halo_arrays["index"][todo[i]])
        # same convention as the reference: report the halo, then abort the chunk
        print(f"Object with HaloCatalogueIndex={index} encountered an error")
        raise RuntimeError(_FATAL_TEXT[int(status[i])].format(index=index))

    ok = status == STATUS_OK
    again = status == STATUS_NEED_LARGER_READ_RADIUS
    upd = again | ok   # finished halos carry the radii of their in-engine retries (resident_radius), if any
    if todo.size == len(halo_arrays["done"]):
        # first call for a chunk: rows of the engine ARE the rows of halo_arrays (masked copies instead of index gathers)
        if ok.all():
            halo_arrays["done"][:] = 1
        else:
            halo_arrays["done"][ok] = 1
        np.copyto(halo_arrays["read_radius"], out["read_radius"], where=upd)
        np.copyto(halo_arrays["search_radius"], out["search_radius"], where=upd)
    else:
        halo_arrays["done"][todo[ok]] = 1
        halo_arrays["read_radius"][todo[upd]] = out["read_radius"][upd]
        halo_arrays["search_radius"][todo[upd]] = out["search_radius"][upd]
    # --record-halo-timings columns (halo_centres.py:205-226): loops of the x1.2 radius search and attempts, summed
    # over every time the halo was processed (halo_tasks.py:188-191 for the attempt that finishes, :411-414 otherwise)
    if "n_loop" in halo_arrays:
        halo_arrays["n_loop"][todo[upd]] += out["stats"][upd, 4].astype(halo_arrays["n_loop"].dtype)
    if "n_process" in halo_arrays:
        halo_arrays["n_process"][todo[upd]] += out["stats"][upd, 5].astype(halo_arrays["n_process"].dtype)

    if unpack:
        coloff = np.concatenate([[0], np.cumsum([engine.desc_columns(c) for c in calcs])])
        table = out["table"]
        for j in np.flatnonzero(ok):
            halo_result = unpack_row(halo_prop_list, coloff, table[j])
            i = todo[j]
            for name, (dset, dtype, desc, phys, aexp) in _INPUT_HALO_COLUMNS.items():
                halo_result[dset] = (np.asarray(halo_arrays[name][i]).astype(dtype), desc, phys, aexp)
            for name, desc in _TIMING_COLUMNS.items():
                if name in halo_arrays:
                    halo_result[f"InputHalos/{name}"] = (np.asarray(halo_arrays[name][i]), desc, True, None)
            results.append(halo_result)

    nr_left = len(halo_arrays["done"]) - int(np.count_nonzero(halo_arrays["done"]))
    raw = dict(out, rows=todo)
    return time.time() - t0_all, task_time, nr_left, int(np.count_nonzero(ok)), raw
