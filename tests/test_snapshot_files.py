"""
The HDF5-free snapshot container and catalogue table (soap_amd/core/snapshot_files.py): what is
written is what is read, cell-masked reads plan contiguous pieces as the reference does (swift_cells.py:480-534), and a box
processed chunk by chunk FROM THE FILES gives the catalogue of the same box processed from memory, bit for bit.
"""

import json
import os

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd.core import chunk_tasks, snapshot_files
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy

DIM = (8, 8, 8)


def _box_on_disk(tmp_path, npart, nhalo, seed, hydro=False, **kw):
    sc, data, halos = make_chunk(npart, nhalo, seed=seed, hydro=hydro, **kw)
    data_np, halos_np = to_numpy(data, halos)
    # the table on disk holds REncloseComoving; the search radius is 1.01 x that, computed by the reader (read_hbtplus.py:319-321)
    r_enclose = halos_np["search_radius"] / 1.01
    halos_np["search_radius"] = 1.01 * r_enclose
    rng = np.random.default_rng(seed)
    shuffled = {}
    for pt, d in data_np.items():   # a snapshot the exporter has to sort: particles in no particular order
        perm = rng.permutation(d["Coordinates"].shape[0])
        shuffled[pt] = {k: v[perm] for k, v in d.items()}
    path = str(tmp_path / "snap")
    snapshot_files.write_snapshot(path, shuffled, sc, DIM, extra_header={"snapshot": "synthetic"})
    snapshot_files.write_catalogue(path, halos_np["index"], halos_np["cofp"], r_enclose, halos_np["is_central"], halos_np["nr_bound_part"])
    return sc, shuffled, halos_np, path


def test_container_round_trip_and_masked_reads(tmp_path):
    sc, shuffled, halos_np, path = _box_on_disk(tmp_path, 120000, 60, seed=5, hydro=True)
    mem = chunk_tasks.SnapshotCells.from_unsorted(shuffled, sc.boxsize, DIM)
    snap = snapshot_files.SnapshotFiles(path, pinned=False)
    assert snap.dimension == DIM and snap.boxsize == sc.boxsize and snap.nr_particles() == mem.nr_particles()
    assert snap.scalars() == sc and snap.header["snapshot"] == "synthetic"
    assert set(snap.data) == set(mem.data)
    for pt in mem.data:
        assert np.array_equal(snap.cell_offset[pt], mem.cell_offset[pt])
        for k, v in mem.data[pt].items():
            assert isinstance(snap.data[pt][k], np.memmap) and snap.data[pt][k].dtype == v.dtype
            assert np.array_equal(np.asarray(snap.data[pt][k]), v)
    rng = np.random.default_rng(1)
    for frac in (0.0, 0.07, 0.5, 1.0):
        mask = rng.random(DIM) < frac
        got, want = snap.read_masked_cells(mask), mem.read_masked_cells(mask)
        for pt in want:
            for k, v in want[pt].items():
                assert got[pt][k].dtype == v.dtype and np.array_equal(got[pt][k], v), (frac, pt, k)
            # adjacent flagged cells are one piece (swift_cells.py:498-519)
            p0, pn = snap.plan_read(pt, mask)
            flagged = int(np.count_nonzero((np.diff(mem.cell_offset[pt]) > 0) & mask.reshape(-1)))
            assert pn.sum() == want[pt]["Coordinates"].shape[0] and len(p0) <= flagged
            assert np.all(p0[1:] > p0[:-1] + pn[:-1]) if len(p0) > 1 else True
    mask = np.zeros(DIM, dtype=bool)
    mask[2, 3, 1:6] = True   # five cells consecutive in file order: one read per dataset
    for pt in mem.data:
        if np.all(np.diff(mem.cell_offset[pt]).reshape(DIM)[2, 3, 1:6] > 0):
            assert len(snap.plan_read(pt, mask)[0]) == 1
    # the halo table: search radius = 1.01 x REncloseComoving (read_hbtplus.py:319-321), read radius at least the minimum
    halos = snapshot_files.read_catalogue(path)
    for k in ("index", "cofp", "is_central", "nr_bound_part"):
        assert np.array_equal(halos[k], halos_np[k]) and halos[k].dtype == halos_np[k].dtype, k
    assert np.array_equal(halos["search_radius"], halos_np["search_radius"])
    assert np.array_equal(halos["read_radius"], halos["search_radius"])
    floor = float(np.median(halos["search_radius"]))
    assert np.array_equal(snapshot_files.read_catalogue(path, floor)["read_radius"], np.maximum(halos["search_radius"], floor))


def test_container_refuses_files_that_do_not_match_their_index(tmp_path):
    sc, shuffled, halos_np, path = _box_on_disk(tmp_path, 30000, 10, seed=6)
    fn = os.path.join(path, "PartType1.Masses.bin")
    with open(fn, "r+b") as f:
        f.truncate(os.path.getsize(fn) - 8)
    with pytest.raises(ValueError, match="size on disk"):
        snapshot_files.SnapshotFiles(path, pinned=False)
    with open(os.path.join(path, "index.json")) as f:
        index = json.load(f)
    index["format"] = "something else"
    with open(os.path.join(path, "index.json"), "w") as f:
        json.dump(index, f)
    with pytest.raises(ValueError, match="not a soap_amd.snapshot"):
        snapshot_files.SnapshotFiles(path, pinned=False)
    with pytest.raises(ValueError, match="datasets missing"):
        snapshot_files.write_snapshot(str(tmp_path / "bad"), {1: {"Coordinates": np.zeros((4, 3))}}, sc, DIM)


@pytest.mark.gpu
def test_chunks_read_from_files_give_the_catalogue_of_the_in_memory_run(built_library, tmp_path):
    """1e6 particles written to disk, then the reference's chunk loop (chunk_tasks.py:189-367) over 3 chunks reading cells from the
    files -- with halos that come back asking for a larger read radius -- against the same loop over the arrays in memory"""
    from soap_amd.core import halo_tasks
    from soap_amd.core.result_set import ResultSet

    sc, shuffled, halos_np, path = _box_on_disk(tmp_path, 1000000, 600, seed=8, read_radius_factor=1.5, satellite_fraction=0.1)
    snap = snapshot_files.SnapshotFiles(path)
    mem = chunk_tasks.SnapshotCells.from_unsorted(shuffled, sc.boxsize, DIM)
    sc_file = snap.scalars()
    assert sc_file == sc
    halos_file = snapshot_files.read_catalogue(path)
    # (the synthetic catalogue's read radius is 1.5 x its search radius; a table on disk starts from the search radius)
    halos_file["read_radius"] = halos_np["read_radius"].copy()
    assert np.array_equal(halos_file["search_radius"], halos_np["search_radius"])
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc_file, cf, None, chp.aperture_variations((30.0, 50.0, 100.0)))

    def run(snapshot, halos):
        engine = halo_tasks.make_engine(sc_file, device=0)
        rs = ResultSet(props, engine.desc_columns)
        logs, raw = [], []
        for part in chunk_tasks.split_halos(halos, sc.boxsize, 3):
            log = []
            chunk_tasks.run_chunk_task(engine, snapshot, props, sc_file, part, results=rs, log=log, raw_tables=raw)
            assert np.all(part["done"] == 1)
            logs.append(log)
        engine.close()
        return rs.columns(), logs, raw

    cols_f, logs_f, raw_f = run(snap, halos_file)
    cols_m, logs_m, raw_m = run(mem, {k: v.copy() for k, v in halos_np.items()})
    assert logs_f == logs_m
    assert max(len(lg) for lg in logs_f) > 1, "no halo asked for a larger read radius: the retry path was not exercised"
    assert all(lg[0]["particles"] < mem.nr_particles() for lg in logs_f)   # partial reads
    assert set(cols_f) == set(cols_m) and len(cols_f["InputHalos/HaloCatalogueIndex"]) == len(halos_np["index"])
    for name in cols_m:
        assert cols_f[name].dtype == cols_m[name].dtype and np.array_equal(cols_f[name], cols_m[name]), name
    for a, b in zip(raw_f, raw_m):
        # (statistics: radius rounds, sphere content, candidates, rounds and attempts of all tries; column 3 is a clock)
        assert np.array_equal(a["table"], b["table"]) and np.array_equal(a["stats"][:, [0, 1, 2, 4, 5]], b["stats"][:, [0, 1, 2, 4, 5]])
    assert snap.bytes_read > 0 and snap.pieces_read > 0
