"""
BASELINE config 4 on one GPU: the SAME box split into chunks (Peano-Hilbert halo chunks, every chunk reads the
top-level cells its halos' read spheres overlap -- boundary particles replicated), the chunks
processed one after the other, their catalogues concatenated with `gather_catalogue`; the result must be the
unsplit run of the same box in (cell, catalogue index) order.

Bit-for-bit when the coordinates are left as stored (`wrap=False`) and every halo runs in the same thread-group
class in both runs: then a halo sees the same particles in the same radial order and the same reduction tree.
(The class is picked from a work estimate that depends on the chunk's mesh, and with `wrap=True` the
reference's box_wrap, chunk_tasks.py:48-50, re-rounds the coordinates about a chunk-dependent reference point:
either moves sums in the last bits, for the reference's own chunked runs as well.  Those runs are held to
1e-10 with exact counts.)
"""

import numpy as np
import pytest
import torch

from soap_amd import compute_halo_properties as chp
from soap_amd import layout
from soap_amd.catalogue import gather_catalogue
from soap_amd.core import chunk_tasks, halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.synthetic import make_chunk, to_numpy

pytestmark = pytest.mark.gpu

DIM = (16, 16, 16)


def _catalogue(raw_tables, cell_size):
    cat = {k: np.concatenate([r[k] for r in raw_tables], axis=0) for k in raw_tables[0]}
    t, c, i = gather_catalogue(torch.from_numpy(cat["table"]), torch.from_numpy(cat["cofp"]),
                               torch.from_numpy(cat["index"]), cell_size=cell_size, dimension=DIM)  # fmt: skip
    order = {int(ix): k for k, ix in enumerate(cat["index"])}
    rows = np.array([order[int(ix)] for ix in i.numpy()])
    return dict(table=t.numpy(), cofp=c.numpy(), index=i.numpy(), stats=cat["stats"][rows],
                search_radius=cat["search_radius"][rows], nr_bound_part=cat["nr_bound_part"][rows])


def _run(sc, snap, props, halos_np, nr_chunks, wrap, **engine_kw):
    engine = halo_tasks.make_engine(sc, device=0, **engine_kw)
    raw, logs = [], []
    for part in chunk_tasks.split_halos(halos_np, sc.boxsize, nr_chunks):
        log = []
        chunk_tasks.run_chunk_task(engine, snap, props, sc, part, wrap=wrap, raw_tables=raw, log=log)
        assert np.all(part["done"] == 1)
        logs.append(log)
    engine.close()
    return _catalogue(raw, snap.cell_size), logs


def _count_columns(props, engine_cols):
    """table columns that hold particle counts, for every calculation of the list"""
    cols, off = [], 0
    for i, hp in enumerate(props):
        names = layout.columns_of(hp.kind)
        nblk = 3 if hp.kind == layout.CALC_PROJECTED else 1
        for b in range(nblk):
            for key in ("Ngas", "Ndm", "Nstar", "Nbh"):
                if hp.kind == layout.CALC_PROJECTED:
                    cols.append(off + layout.projected_columns(key, b)[0])
                elif key in names:
                    cols.append(off + names[key][0])
        off += engine_cols[i]
    return np.array(cols)


@pytest.mark.parametrize("nr_chunks", [8, 3])
def test_union_of_chunk_catalogues_is_the_unsplit_catalogue(built_library, nr_chunks):
    sc, data, halos = make_chunk(400000, 320, seed=21, satellite_fraction=0.1)
    data_np, halos_np = to_numpy(data, halos)
    snap = chunk_tasks.SnapshotCells(data_np, sc.boxsize, DIM)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 50.0, 100.0)))
    # one thread-group class for every halo, in both runs (see the module docstring)
    kw = dict(slab_capacity=1 << 18, n_workgroups=8)
    whole, log1 = _run(sc, snap, props, halos_np, 1, wrap=False, **kw)
    split, logn = _run(sc, snap, props, halos_np, nr_chunks, wrap=False, **kw)
    assert len(whole["index"]) == len(halos_np["index"]) == len(split["index"])
    assert np.array_equal(whole["index"], split["index"])
    assert np.array_equal(whole["cofp"], split["cofp"])
    assert np.array_equal(whole["nr_bound_part"], split["nr_bound_part"])
    # every column of every halo, bit for bit (NaN-free by construction)
    assert np.array_equal(whole["table"], split["table"])
    # same number of radius iterations and the same final search sphere; the candidate count (column 2) belongs to
    # the chunk's own mesh and is not comparable
    assert np.array_equal(whole["stats"][:, :2], split["stats"][:, :2])
    assert np.array_equal(whole["search_radius"], split["search_radius"])
    # the chunks really were partial reads, and together they read more than the box (replicated boundary cells)
    n_all = snap.nr_particles()
    first_reads = [lg[0]["particles"] for lg in logn]
    assert max(first_reads) < n_all and sum(first_reads) > n_all
    assert max(first_reads) < log1[0][0]["particles"] <= n_all


def test_split_with_box_wrap_and_default_classes(built_library):
    sc, data, halos = make_chunk(400000, 320, seed=22)
    data_np, halos_np = to_numpy(data, halos)
    snap = chunk_tasks.SnapshotCells(data_np, sc.boxsize, DIM)
    cf = CategoryFilter(dmo=True)
    props = chp.build_halo_prop_list(sc, cf, None, chp.aperture_variations((30.0, 50.0, 100.0)))
    whole, _ = _run(sc, snap, props, halos_np, 1, wrap=False)
    split, _ = _run(sc, snap, props, halos_np, 8, wrap=True)
    assert np.array_equal(whole["index"], split["index"])
    assert np.array_equal(whole["stats"][:, :2], split["stats"][:, :2])
    engine = halo_tasks.make_engine(sc, device=0)
    ncols = [engine.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)]
    engine.close()
    cc = _count_columns(props, ncols)
    assert np.array_equal(whole["table"][:, cc], split["table"][:, cc])
    a, b = whole["table"], split["table"]
    scale = np.maximum(np.abs(a).max(axis=0, keepdims=True), 1e-300)   # per column: sums of signed terms cancel
    # 1e-10 is the bar of the north star; the wrap re-rounds coordinates by ~1e-16 L, which sums of signed terms
    # (angular momenta, off-diagonal dispersions) amplify to ~1e-11 of their value
    assert np.all(np.abs(a - b) <= 1e-10 * np.abs(a) + 1e-12 * scale)
