"""
N>1 path on CPU: two gloo ranks, each with its own part of the halo table, gather the
catalogue and must end up with the same, correctly ordered concatenation
(SOAP/core/combine_chunks.py:33-61 ordering rule).
"""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from soap_amd.catalogue import gather_catalogue, spatial_sort_order
    from soap_amd.core.chunk_tasks import split_halos

    rng = np.random.default_rng(99)
    n, ncols, L = 137, 11, 100.0
    cofp = rng.uniform(0, L, (n, 3))
    index = rng.permutation(n).astype(np.int64) * 3 + 1
    table = rng.normal(size=(n, ncols))
    # the product's own split: Peano-Hilbert chunks of the halo table (chunk_tasks.split_halos, what bench.py --gpus N
    # and run_chunk_task use), every rank takes its chunk
    halos = dict(index=index, cofp=cofp, nr_bound_part=rng.integers(20, 5000, n), row=np.arange(n))
    chunks = split_halos(halos, L, world)
    assert sum(len(c["index"]) for c in chunks) == n and abs(len(chunks[0]["index"]) - len(chunks[1]["index"])) <= 1
    mine = np.zeros(n, dtype=bool)
    mine[chunks[rank]["row"]] = True
    assert 0 < mine.sum() < n
    t, c, i = gather_catalogue(torch.from_numpy(table[mine]), torch.from_numpy(cofp[mine]),
                               torch.from_numpy(index[mine]), cell_size=L / 4, dimension=(4, 4, 4))  # fmt: skip
    order = spatial_sort_order(cofp, index, L / 4, (4, 4, 4))
    assert np.array_equal(i.numpy(), index[order])
    assert np.array_equal(c.numpy(), cofp[order])
    assert np.array_equal(t.numpy(), table[order])
    # the stored catalogue (float32 datasets + float64 / integer datasets) travels the same way
    from soap_amd.catalogue import gather_catalogue_packed

    narrow = rng.normal(size=(n, 7)).astype(np.float32)
    n32, w64, c2, i2 = gather_catalogue_packed(torch.from_numpy(narrow[mine]), torch.from_numpy(table[mine]), torch.from_numpy(cofp[mine]),
                                               torch.from_numpy(index[mine]), cell_size=L / 4, dimension=(4, 4, 4))  # fmt: skip
    assert n32.dtype == torch.float32 and np.array_equal(n32.numpy(), narrow[order]) and np.array_equal(w64.numpy(), table[order])
    assert np.array_equal(i2.numpy(), index[order]) and np.array_equal(c2.numpy(), cofp[order])
    np.save(os.path.join(tmpdir, f"ok{rank}.npy"), i.numpy())
    dist.destroy_process_group()


def test_catalogue_gather_two_ranks(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "ok0.npy")
    b = np.load(tmp_path / "ok1.npy")
    assert np.array_equal(a, b) and len(a) == 137


def test_split_is_balanced_and_compact():
    """the split the ranks use (Peano-Hilbert keys, equal counts: domain_decomposition.py:64-142)"""
    sys.path.insert(0, ROOT)
    from soap_amd.core.chunk_tasks import split_halos

    rng = np.random.default_rng(5)
    cofp = rng.uniform(0, 50.0, (8000, 3))
    halos = dict(index=np.arange(8000, dtype=np.int64), cofp=cofp, nr_bound_part=rng.integers(20, 5000, 8000))
    chunks = split_halos(halos, 50.0, 8)
    counts = np.array([len(c["index"]) for c in chunks])
    assert counts.min() == counts.max() == 1000
    assert np.array_equal(np.sort(np.concatenate([c["index"] for c in chunks])), np.arange(8000))
    # compact: the mean extent of a chunk is well below the box size
    ext = np.mean([np.ptp(c["cofp"], axis=0).mean() for c in chunks])
    assert ext < 0.8 * 50.0
    # inside a chunk: most bound particles first (chunk_tasks.py:118-120)
    assert all(np.all(np.diff(c["nr_bound_part"]) <= 0) for c in chunks)
