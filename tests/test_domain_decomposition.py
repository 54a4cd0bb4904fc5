"""
Host logic of the split of one box into chunks (CPU): Peano-Hilbert keys, equal-count chunks,
`separate_chunks`, top-level-cell masks and the cell-masked read
(SOAP/core/domain_decomposition.py:9-142, mask_cells.py:6-38, swift_cells.py:540-549).
"""

import numpy as np
import pytest

from soap_amd.core import chunk_tasks
from soap_amd.core.domain_decomposition import peano_decomposition, peano_hilbert_keys
from soap_amd.synthetic import make_chunk, to_numpy


@pytest.mark.parametrize("bits", [1, 2, 3, 4])
def test_peano_hilbert_key_is_a_hilbert_curve(bits):
    """The defining properties of the Gadget-2 / virgo key: a bijection of the 2^bits cube onto
    0 .. 8^bits - 1 whose consecutive cells are face neighbours, starting in the corner cell (0, 0, 0)."""
    n = 1 << bits
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    key = peano_hilbert_keys(g[:, 0], g[:, 1], g[:, 2], bits)
    assert np.array_equal(np.sort(key), np.arange(n**3))
    walk = g[np.argsort(key)]
    assert np.all(np.abs(np.diff(walk, axis=0)).sum(axis=1) == 1)
    assert tuple(walk[0]) == (0, 0, 0)


def test_peano_keys_refine_consistently():
    """The first 3 k bits of a key are the key of the parent cell at level k (what makes chunks of consecutive
    keys spatially compact)."""
    rng = np.random.default_rng(3)
    p = rng.integers(0, 1024, (2000, 3))
    k10 = peano_hilbert_keys(p[:, 0], p[:, 1], p[:, 2], 10)
    k4 = peano_hilbert_keys(p[:, 0] >> 6, p[:, 1] >> 6, p[:, 2] >> 6, 4)
    assert np.array_equal(k10 >> 18, k4)


def test_decomposition_chunk_sizes_and_order():
    rng = np.random.default_rng(5)
    n, L = 1003, 50.0
    halo = dict(cofp=rng.uniform(0, L, (n, 3)), index=np.arange(n), nr_bound_part=rng.integers(20, 5000, n))
    out, chunk_size = peano_decomposition(L, halo, 8)
    # domain_decomposition.py:135-140: total // n each, the first total % n chunks one more
    assert list(chunk_size) == [126, 126, 126, 125, 125, 125, 125, 125]
    assert sorted(out["index"]) == list(range(n))
    ip = np.clip(np.floor(out["cofp"] / (L / 1024)).astype(np.int64), 0, 1023)
    keys = peano_hilbert_keys(ip[:, 0], ip[:, 1], ip[:, 2], 10)
    assert np.all(np.diff(keys) >= 0)
    # compact: a chunk's extent is well below the box
    first = np.cumsum(chunk_size) - chunk_size
    ext = np.mean([np.ptp(out["cofp"][a : a + s], axis=0).mean() for a, s in zip(first, chunk_size)])
    assert ext < 0.75 * L
    # fewer halos than chunks: every chunk keeps at least one halo (:75-77)
    small = {k: v[:3] for k, v in halo.items()}
    _, cs = peano_decomposition(L, small, 8)
    assert list(cs) == [1, 1, 1]


def test_separate_chunks_take_the_giants_out_of_the_curve():
    rng = np.random.default_rng(6)
    n, L = 200, 80.0
    nb = rng.integers(20, 900, n)
    nb[[5, 17, 99]] = [50000, 80000, 20000]
    halo = dict(cofp=rng.uniform(0, L, (n, 3)), index=np.arange(n), nr_bound_part=nb)
    sep = [dict(n_bound_threshold=60000, n_halo_per_chunk=1), dict(n_bound_threshold=10000, n_halo_per_chunk=2)]
    out, chunk_size = peano_decomposition(L, halo, 4, separate_chunks=sep)
    # 197 small halos on 4 curve chunks, then [80000] alone and [50000, 20000] together (:96-133)
    assert list(chunk_size) == [50, 49, 49, 49, 1, 2]
    assert list(out["index"][-3:]) == [17, 5, 99]


def test_mask_cells_matches_the_reference_loops():
    rng = np.random.default_rng(1)
    dim, L = (8, 8, 8), 100.0
    cs = L / 8
    c = rng.uniform(0, L, (60, 3))
    r = rng.uniform(0.5, 30, 60)
    r[7] = 70.0   # a read sphere wider than the box flags every cell once
    done = (rng.uniform(size=60) < 0.3).astype(int)
    done[7] = 1
    for trial in range(2):
        m = chunk_tasks.mask_cells(cs, dim, c, r, done)
        ref = np.zeros(dim, bool)
        for h in range(60):
            if done[h]:
                continue
            imin = np.floor((c[h] - r[h] - 0.5 * cs) / cs).astype(int)
            imax = np.floor((c[h] + r[h] + 0.5 * cs) / cs).astype(int)
            for i in range(imin[0], imax[0] + 1):
                for j in range(imin[1], imax[1] + 1):
                    for k in range(imin[2], imax[2] + 1):
                        ref[i % 8, j % 8, k % 8] = True
        assert np.array_equal(m, ref)
        done[7] = 0
    assert m.all()


def test_cell_masked_read_returns_every_particle_inside_the_read_spheres():
    sc, data, halos = make_chunk(120000, 120, seed=8)
    d, h = to_numpy(data, halos)
    L = sc.boxsize
    snap = chunk_tasks.SnapshotCells(d, L, (16, 16, 16))
    assert snap.nr_particles() == 120000
    parts = chunk_tasks.split_halos(h, L, 4)
    assert sorted(np.concatenate([p["index"] for p in parts])) == sorted(h["index"])
    pos = d[1]["Coordinates"]
    total = 0
    for p in parts:
        assert np.all(np.diff(p["nr_bound_part"]) <= 0)   # biggest first (chunk_tasks.py:118)
        got, ncells = chunk_tasks.read_chunk(snap, p, wrap=False)
        gp = got[1]["Coordinates"]
        total += gp.shape[0]
        have = set(map(bytes, gp))
        for i in range(0, len(p["index"]), 7):
            dr = pos - p["cofp"][i]
            dr -= L * np.round(dr / L)
            inside = pos[np.einsum("ij,ij->i", dr, dr) <= p["read_radius"][i] ** 2]
            assert all(bytes(x) in have for x in inside)
        # the wrap moves coordinates by multiples of the box size only and makes the region contiguous
        wrapped, _ = chunk_tasks.read_chunk(snap, p, wrap=True)
        wp = wrapped[1]["Coordinates"]
        k = (wp - gp) / L
        assert np.allclose(k, np.round(k), atol=1e-9)
        ref = chunk_tasks.chunk_ref_pos(p["cofp"])
        assert np.all(np.abs(wp - ref) <= 0.5 * L * (1 + 1e-12))
        for name in ("Velocities", "Masses", "GroupNr_bound", "FOFGroupIDs"):
            assert np.array_equal(wrapped[1][name], got[1][name])
    assert total >= 120000   # boundary cells are replicated, none is lost
    # all halos done: nothing to read
    done = dict(parts[0], done=np.ones(len(parts[0]["index"]), dtype=np.int32))
    assert not chunk_tasks.mask_cells(snap.cell_size, snap.dimension, done["cofp"], done["read_radius"], done["done"]).any()


def test_snapshot_cells_rejects_unsorted_particles_and_sorts_on_request():
    rng = np.random.default_rng(2)
    pos = rng.uniform(0, 10.0, (500, 3))
    data = {1: {"Coordinates": pos, "Masses": np.arange(500.0)}}
    with pytest.raises(ValueError):
        chunk_tasks.SnapshotCells(data, 10.0, (4, 4, 4))
    snap = chunk_tasks.SnapshotCells.from_unsorted(data, 10.0, (4, 4, 4))
    off = snap.cell_offset[1]
    assert off[0] == 0 and off[-1] == 500 and np.all(np.diff(off) >= 0)
    mask = np.zeros((4, 4, 4), bool)
    mask[1, 2, 3] = True
    got = snap.read_masked_cells(mask)[1]
    want = np.all(np.floor(pos / 2.5).astype(int) == [1, 2, 3], axis=1)
    assert sorted(got["Masses"]) == sorted(np.arange(500.0)[want])


def test_weighted_cut_balances_total_weight_and_keeps_the_curve_order():
    """`weights`: same Peano-Hilbert order, boundaries at equal cumulative weight; without weights the reference's equal-count cut"""
    from soap_amd.core.domain_decomposition import halo_work_weights

    rng = np.random.default_rng(8)
    n, L, nc = 5000, 80.0, 8
    halos = {"cofp": rng.uniform(0, L, (n, 3)), "index": np.arange(n, dtype=np.int64),
             "nr_bound_part": np.maximum(20, (20.0 * rng.pareto(0.9, n)).astype(np.int64))}  # fmt: skip
    w = halo_work_weights(halos)
    assert np.array_equal(w, halos["nr_bound_part"] + 64.0)
    plain, size0 = peano_decomposition(L, halos, nc)
    weighted, size1 = peano_decomposition(L, halos, nc, weights=w)
    assert "_weight" not in weighted and np.array_equal(plain["index"], weighted["index"])   # same order along the curve
    assert size0.max() - size0.min() <= 1 and size1.sum() == n and np.all(size1 > 0)
    wo = halo_work_weights(weighted)
    share = lambda size: np.array([wo[a:b].sum() for a, b in zip(np.cumsum(size) - size, np.cumsum(size))])   # noqa: E731
    s0, s1 = share(size0), share(size1)
    # no chunk is heavier than the ideal share plus one halo
    assert s1.max() <= wo.sum() / nc + wo.max() + 1e-9
    assert s1.max() < s0.max()
    # split_halos hands the chunks out, heaviest halo first inside each
    parts = chunk_tasks.split_halos(halos, L, nc, weights="work")
    assert [len(p["index"]) for p in parts] == list(size1)
    assert all(np.all(np.diff(p["nr_bound_part"]) <= 0) for p in parts)
    # degenerate cases: one giant weight, more chunks than halos
    w2 = np.ones(n)
    w2[n // 2] = 1e9
    _, size2 = peano_decomposition(L, halos, nc, weights=w2)
    assert size2.sum() == n and np.all(size2 > 0)
    few = {k: v[:3] for k, v in halos.items()}
    _, size3 = peano_decomposition(L, few, nc, weights=np.ones(3))
    assert list(size3) == [1, 1, 1]
    with pytest.raises(ValueError):
        peano_decomposition(L, halos, nc, weights=np.zeros(n))
