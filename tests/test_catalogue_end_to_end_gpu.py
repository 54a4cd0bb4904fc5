"""
SURVEY.md section 8 row f1, end to end on the C1-substitute box (1e5 particles in a periodic box): the catalogue the
product writes -- snapshot in top-level-cell order, Peano-Hilbert halo chunks, cell-masked reads with read-radius
retries, engine, ResultSet, (cell, catalogue index) order, writer -- against a catalogue assembled from the ORACLE run
on the unsplit box: same dataset names, dtypes, shapes, row order, units / descriptions, and every value.
"""
import json
import os

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd.core import chunk_tasks, halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.core.result_set import ResultSet
from soap_amd.synthetic import make_chunk, to_numpy
from tests import helpers
from tests.test_engine_contract_gpu import _ulp32
from tests.test_parity_gpu import _abs_scale, _scales

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "property_rows.json")) as _f:
    REF_ROWS = json.load(_f)   # rows of the reference's own property table (tests/golden/make_property_rows.py)


def _a_power(key):
    """power of a between the comoving expression and the stored unit, from the reference's table row alone"""
    rr = REF_ROWS[key]
    if not rr["output_physical"]:
        return 0
    if key in ("Ldm", "Lgas", "Lstar", "Lbaryons"):
        return 1
    if "InertiaTensor" in key and "snap_length" in rr["unit"]:
        return 2
    return -1 if key.endswith("FlowRate") else 0


def _oracle_catalogue(sc, data_np, halos_np, cf, so_vars, ap_vars, pj_vars, dimension):
    """{dataset name: array over halos} from the oracle on the unsplit box, in the reference's final order
    (combine_chunks.py:33-52: cell of the halo centre, then catalogue index)."""
    from oracle import halo_oracle as ho

    u, props = helpers.oracle_prop_list(sc, cf, so_vars, ap_vars, pj_vars)
    mesh = {pt: ho.Mesh(d["Coordinates"], ho.mesh_resolution(d["Coordinates"].shape[0])) for pt, d in data_np.items()}
    ha = dict({k: v.copy() for k, v in halos_np.items()}, done=np.zeros(len(halos_np["index"]), dtype=np.int32))
    rows = {}
    for _ in range(12):
        if not np.any(ha["done"] == 0):
            break
        rows.update(dict(ho.process_halos(mesh, data_np, props, u, ha)))
    assert np.all(ha["done"] == 1)
    H = len(ha["index"])
    cols, raws = {}, {}
    dmo = cf.dmo
    for okey in rows[0]:
        group, key = okey.rsplit("/", 1)
        rr = REF_ROWS[key]
        if dmo and not rr["dmo_property"]:
            continue   # SO_properties.py:3607-3608
        k = _a_power(key)
        out = np.zeros((H, rr["shape"]) if rr["shape"] > 1 else (H,), dtype=np.dtype(rr["dtype"]))
        raw = np.stack([rows[i][okey].raw for i in range(H)])
        if np.issubdtype(out.dtype, np.integer):
            out[...] = raw.reshape(out.shape).astype(out.dtype)
        else:
            out += (raw * sc.a**k if k else raw).reshape(out.shape)   # zero-initialised table-dtype array += value
        cols[f"{group}/{rr['name']}"] = out
        raws[f"{group}/{rr['name']}"] = (key, group, raw * sc.a**k)
    cols["InputHalos/HaloCentre"] = halos_np["cofp"].astype(np.float64)
    cols["InputHalos/HaloCatalogueIndex"] = halos_np["index"].astype(np.int64)
    cols["InputHalos/IsCentral"] = halos_np["is_central"].astype(np.int64)
    cols["InputHalos/NumberOfBoundParticles"] = halos_np["nr_bound_part"].astype(np.int64)
    cell_size = sc.boxsize / np.asarray(dimension, dtype=np.float64)
    ci = (halos_np["cofp"] // cell_size).astype(np.int64)
    cell_index = ci[:, 0] * dimension[0] ** 2 + ci[:, 1] * dimension[1] + ci[:, 2]
    order = np.lexsort((halos_np["index"], cell_index))
    return {k: v[order] for k, v in cols.items()}, {k: (a, b, c[order]) for k, (a, b, c) in raws.items()}, [rows[i] for i in order]


@pytest.mark.parametrize("hydro,nr_chunks", [(False, 5), (True, 3)])
def test_written_catalogue_equals_the_oracle_catalogue(built_library, tmp_path, hydro, nr_chunks):
    sc, data, halos = make_chunk(100000, 110, seed=2024, hydro=hydro, satellite_fraction=0.12, read_radius_factor=1.2)
    data_np, halos_np = to_numpy(data, halos)
    cf = CategoryFilter(dmo=not hydro)
    so_vars = None   # the default nine SO variations, incl. BN98, a physical radius and 5xR_500_crit
    ap_vars = chp.aperture_variations((30.0, 50.0, 100.0))
    pj_vars = chp.projected_variations((30.0, 100.0)) if hydro else {}
    dimension = (6, 6, 6)

    # ---- product: the whole chain --------------------------------------------------------------------------------
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars)
    snapshot = chunk_tasks.SnapshotCells.from_unsorted(data_np, sc.boxsize, dimension)
    engine = halo_tasks.make_engine(sc, device=0)
    rs = ResultSet(props, engine.desc_columns)
    n_iterations = 0
    for chunk_halos in chunk_tasks.split_halos(halos_np, sc.boxsize, nr_chunks):
        log = []
        chunk_tasks.run_chunk_task(engine, snapshot, props, sc, chunk_halos, wrap=True, results=rs, log=log)
        assert np.all(chunk_halos["done"] == 1)
        n_iterations += len(log)
    # the multi-GPU route of the same catalogue: every chunk's rows packed on the device (stored dtypes), concatenated in
    # final order by gather_catalogue_packed, columns cut on the host -- must be the ResultSet above, bit for bit
    import torch

    from soap_amd.catalogue import gather_catalogue_packed
    from soap_amd.core.result_set import PackedLayout

    packed = PackedLayout(props, engine.desc_columns)
    parts = []
    for chunk_halos in chunk_tasks.split_halos(halos_np, sc.boxsize, nr_chunks):
        raw_tables = []
        chunk_tasks.run_chunk_task(engine, snapshot, props, sc, chunk_halos, wrap=True, raw_tables=raw_tables)
        for rt in raw_tables:
            narrow, wide = packed.pack(torch.as_tensor(rt["table"], device="cuda:0"))
            parts.append((narrow, wide, rt["cofp"], rt["index"], rt["is_central"], rt["nr_bound_part"]))
    narrow = torch.cat([p[0] for p in parts]); wide = torch.cat([p[1] for p in parts])
    cofp_all = np.concatenate([p[2] for p in parts]); index_all = np.concatenate([p[3] for p in parts])
    n_all, w_all, c_all, i_all = gather_catalogue_packed(narrow, wide, torch.as_tensor(cofp_all, device="cuda:0"),
                                                        torch.as_tensor(index_all, device="cuda:0"), list(snapshot.cell_size), dimension)
    by_index = {int(ix): k for k, ix in enumerate(index_all)}
    perm = np.array([by_index[int(ix)] for ix in i_all.cpu().numpy()])
    rs_packed = ResultSet(props, engine.desc_columns)
    rs_packed.append_packed(packed, n_all.cpu().numpy(), w_all.cpu().numpy(),
                            dict(cofp=c_all.cpu().numpy(), index=i_all.cpu().numpy(), is_central=np.concatenate([p[4] for p in parts])[perm],
                                 nr_bound_part=np.concatenate([p[5] for p in parts])[perm]))
    engine.close()
    ref_cols = rs.sorted_columns(list(snapshot.cell_size), list(dimension))
    pk_cols = rs_packed.columns()
    assert set(ref_cols) == set(pk_cols)
    for k in ref_cols:
        assert ref_cols[k].dtype == pk_cols[k].dtype and np.array_equal(ref_cols[k], pk_cols[k], equal_nan=True), k
    assert n_iterations > nr_chunks, "the box should need at least one read-radius retry"
    path = str(tmp_path / "catalogue")
    rs.write(path, cell_size=list(snapshot.cell_size), dimension=list(dimension))
    z = np.load(path + ".npz")
    got = {k.replace("|", "/"): z[k] for k in z.files}
    with open(path + ".json") as f:
        meta = json.load(f)

    # ---- oracle catalogue on the unsplit box ---------------------------------------------------------------------
    want, raws, orows = _oracle_catalogue(sc, data_np, halos_np, cf, so_vars, ap_vars, pj_vars, dimension)

    assert set(got) == set(want), (sorted(set(got) ^ set(want))[:10])
    assert np.array_equal(got["InputHalos/HaloCatalogueIndex"], want["InputHalos/HaloCatalogueIndex"])   # row order
    by_name = {rr["name"]: rr for rr in REF_ROWS.values()}
    n32 = nexact = 0
    for name, w in want.items():
        g = got[name]
        assert g.dtype == w.dtype and g.shape == w.shape, (name, g.dtype, w.dtype, g.shape, w.shape)
        if name.startswith("InputHalos/"):
            assert np.array_equal(g, w), name
            continue
        rr = by_name[name.rsplit("/", 1)[1]]
        m = meta[name]
        assert m["unit"] == rr["unit"] and m["output_physical"] == rr["output_physical"], (name, m)
        assert m["a_scale_exponent"] == rr["a_scale_exponent"] and m["dtype"] == rr["dtype"], (name, m)
        if np.issubdtype(w.dtype, np.integer):
            assert np.array_equal(g, w), name
            continue
        if w.dtype == np.float64:
            assert np.allclose(g, w, rtol=1e-10, atol=1e-10 * sc.boxsize), name
            continue
        key, group, raw = raws[name]
        cancelling = key.startswith(("L", "veldisp", "vcom", "com", "proj_veldisp", "kappa", "DtoT")) or "Inertia" in key
        for row in range(len(w)):
            tol = _ulp32(w[row])
            if cancelling:
                M, vscale = _scales(orows[row], group)
                tol = tol + 1e-9 * float(np.max(np.abs(raw[row]))) + 1e-12 * _abs_scale(key, M, vscale) * sc.a ** _a_power(key)
            d = np.abs(np.atleast_1d(g[row]).astype(np.float64) - np.atleast_1d(w[row]).astype(np.float64))
            assert np.all(d <= tol), (name, row, g[row], w[row])
        n32 += w.size
        nexact += int(np.sum(g == w))
    assert n32 > 20000 and nexact > 0.98 * n32, (n32, nexact)
