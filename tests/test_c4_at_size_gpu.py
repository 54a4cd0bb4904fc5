"""
BASELINE config 4 AT SIZE: the seed-1234 C3 box (1e8 particles, 1e5 halos, 16 calculations) cut into 8 Peano-Hilbert halo
chunks as `bench.py --gpus 8` cuts it (SOAP/core/domain_decomposition.py:64-142 through soap_amd.core.chunk_tasks.split_halos);
ranks 0 and 7 are run here, on the one GPU of the test box, exactly as a rank of the 8-GPU job runs them -- cell-masked read
of the top-level cells the rank's read spheres overlap (mask_cells.py:6-38), coordinates wrapped about the chunk
(chunk_tasks.py:48-50, `wrap=True`), resident radii from the mask, the chunk loop with read-radius retries -- and every row of
both ranks is compared with the same halo's row of the UNSPLIT run of the whole box: statuses, radius rounds, sphere contents,
particle counts and radii exactly, values to 1e-10 (the wrap re-rounds coordinates about a chunk-dependent point, see
tests/test_split_box_gpu.py).  What this leaves to the hardware run is the RCCL gather of the eight catalogues.
"""

import time

import numpy as np
import pytest

from soap_amd import layout
from soap_amd.core import chunk_tasks, halo_tasks
from soap_amd.synthetic import make_chunk
from tests.test_split_box_gpu import _count_columns

pytestmark = pytest.mark.gpu


def cut_ranks_equal_the_unsplit_box(sc, data, halos, props, ranks, tag, region_share=(0.1, 0.4)):
    """Run the whole box unsplit, then the given ranks of its 8-way Peano-Hilbert cut as a rank of `bench.py --gpus 8` runs them, and
    compare every row of those ranks with the same halo's row of the unsplit run (shared with tests/test_c5_cut_gpu.py)."""
    import torch

    import bench

    t0 = time.time()
    halos_all = {k: v.cpu().numpy() for k, v in halos.items()}
    H = len(halos_all["index"])
    engine = halo_tasks.make_engine(sc, device=0)
    ncols = [engine.desc_columns(hp.calc_desc(props, i)) for i, hp in enumerate(props)]
    cc = _count_columns(props, ncols)

    # ---- the unsplit box: one chunk, everything resident
    halo_tasks.stage_chunk(engine, data)
    ha = dict({k: v.copy() for k, v in halos_all.items()}, done=np.zeros(H, dtype=np.int32))
    out = halo_tasks.process_halos(engine, props, sc, ha, None, unpack=False, resident_radius=np.full(H, np.inf))
    assert out[2] == 0 and np.all(out[4]["status"] == 0)
    whole = engine.fetch(table=True)
    whole_search, whole_read = ha["search_radius"].copy(), ha["read_radius"].copy()
    row_of = {int(ix): r for r, ix in enumerate(halos_all["index"])}
    print(f"[{tag}] unsplit pass: {time.time() - t0:.1f} s", flush=True)

    # ---- the ranks of the 8-way cut, as a rank of `bench.py --gpus 8` runs them
    snapshot = chunk_tasks.SnapshotCells(data, sc.boxsize, (bench.TOPCELLS,) * 3)
    n_box = snapshot.nr_particles()
    chunks = chunk_tasks.split_halos(halos_all, sc.boxsize, 8)
    assert sum(len(c["index"]) for c in chunks) == H and max(len(c["index"]) for c in chunks) - min(len(c["index"]) for c in chunks) <= 1
    seen = 0
    for rank in ranks:
        hn = chunks[rank]
        Hr = len(hn["index"])
        region, _, mask0 = chunk_tasks.read_chunk(snapshot, hn, wrap=True, return_mask=True)
        n_region = sum(int(d["Coordinates"].shape[0]) for d in region.values())
        assert region_share[0] * n_box < n_region < region_share[1] * n_box, (rank, n_region, n_box)   # a partial read with replicated boundary cells
        resident = chunk_tasks.resident_radius(mask0, snapshot.cell_size, hn["cofp"], hn["read_radius"])
        hr = dict({k: v.copy() for k, v in hn.items()}, done=np.zeros(Hr, dtype=np.int32))
        table = np.zeros((Hr, sum(ncols)))
        stats = np.zeros((Hr, whole["stats"].shape[1]), dtype=np.int64)
        cur = region
        calls = 0
        for it in range(8):
            halo_tasks.stage_chunk(engine, cur)
            o = halo_tasks.process_halos(engine, props, sc, hr, None, unpack=False, resident_radius=resident)
            calls += 1
            raw = o[4]
            got = engine.fetch(table=True)
            ok = raw["status"] == 0
            assert np.all(raw["status"] >= 0)
            table[raw["rows"][ok]] = got["table"][ok]
            stats[raw["rows"][ok]] = raw["stats"][ok]
            if o[2] == 0:
                break
            cur, _, mk = chunk_tasks.read_chunk(snapshot, hr, wrap=True, return_mask=True)
            resident = chunk_tasks.resident_radius(mk, snapshot.cell_size, hr["cofp"], hr["read_radius"])
        assert np.all(hr["done"] == 1), (rank, int(np.sum(hr["done"] == 0)))
        del region, cur
        torch.cuda.empty_cache()
        rows = np.array([row_of[int(ix)] for ix in hn["index"]])
        # radius rounds of the attempt that finished and content of the final sphere; radii handed back (the unsplit run made
        # its read-radius retries inside the engine: the same updates, halo_tasks.py:393-402)
        assert np.array_equal(stats[:, :2], whole["stats"][rows, :2]), rank
        assert np.array_equal(hr["search_radius"], whole_search[rows]) and np.array_equal(hr["read_radius"], whole_read[rows]), rank
        a, b = whole["table"][rows], table
        assert np.array_equal(a[:, cc], b[:, cc]), rank   # particle counts of every calculation
        scale = np.maximum(np.abs(whole["table"]).max(axis=0, keepdims=True), 1e-300)
        rtol = np.full(a.shape, 1e-10)
        # The unsoftened maximum circular velocity of the bound subhalo is max M(<r) / r over the particles, the innermost ones
        # included (subhalo_properties.py:982-1073, kinematic_properties.py:555-593): where it is reached at a radius r the wrap's
        # re-rounding of the coordinates (one ulp of the box size, about the chunk's reference point) moves the value by ulp(L) / r
        # relative -- in the reference's own chunked runs as well.
        sub = layout.columns_of(layout.CALC_SUBHALO)
        r_vmax = np.maximum(np.abs(a[:, sub["R_vmax_unsoft"][0]]), 1e-300)
        for key in ("Vmax_unsoft", "R_vmax_unsoft"):
            rtol[:, sub[key][0]] = np.maximum(1e-10, 8.0 * np.spacing(sc.boxsize) / r_vmax)
        # kappa_corot and the disc fractions (gas / stars, hydro boxes): K_rot,i = L_i^2 / (2 m R_i^2) with R_i^2 = |x|^2 - (x . Lhat)^2 cancels
        # for members near the rotation axis, so the wrap's re-rounded coordinates move the ratio by ~1e-10 relative; the reference holds
        # both in float32 buffers (kinematic_properties.py:360-381): compared at its 2e-7, as in tests/test_parity_gpu.py
        off = 0
        for i, hp in enumerate(props):
            names = layout.columns_of(hp.kind)
            for key, (c0, nc) in names.items():
                if hp.kind != layout.CALC_PROJECTED and (key.startswith("kappa_corot") or key.startswith("DtoT")) and c0 + nc <= ncols[i]:
                    rtol[:, off + c0:off + c0 + nc] = 2e-7
            off += ncols[i]
        # (absolute floor: a length that is exactly zero in one run -- a half-mass radius that is the central particle's own
        # radius -- is a few ulp of the box size in the other, whose wrap has re-rounded that particle's coordinates)
        atol = np.maximum(1e-12 * scale, 8.0 * np.spacing(sc.boxsize))
        bad = np.abs(a - b) > rtol * np.abs(a) + atol
        if bad.any():
            ij = np.argwhere(bad)
            worst = ij[np.argsort(-(np.abs(a - b) / (np.abs(a) + scale))[bad])[:8]]
            raise AssertionError((rank, int(bad.sum()), [(int(i), int(j), float(a[i, j]), float(b[i, j])) for i, j in worst]))
        seen += Hr
        print(f"[{tag}] rank {rank}: {Hr} halos, {n_region} particles ({n_region / n_box:.3f} of the box), {calls} call(s): {time.time() - t0:.1f} s", flush=True)
    assert seen == sum(len(chunks[r]["index"]) for r in ranks)
    engine.close()


def test_ranks_0_and_7_of_the_8_way_split_equal_the_unsplit_c3_box(built_library):
    import bench

    wl = bench.WORKLOADS["c3"]
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", topcells=bench.TOPCELLS)
    props = bench.build_props(sc, wl)[4]
    cut_ranks_equal_the_unsplit_box(sc, data, halos, props, (0, 7), "c4 at size")
