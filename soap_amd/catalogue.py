"""
Final-catalogue concatenation across GPUs: the one exchange step of the path.

The reference writes per-chunk scratch files and then `combine_chunks` does a distributed
sort of all halos by (SWIFT cell of the halo centre, catalogue index)
(SOAP/core/combine_chunks.py:33-61, `spatial_sort`) and `fetch_elements` per property.
Here every rank (one per GPU) holds its chunk's fp64 result table in HBM; ONE all-gather
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests) concatenates the tables
and the halo keys, and every rank applies the same ordering rule locally.  No reduction is
needed: halos are independent.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def spatial_sort_order(halo_cofp: np.ndarray, halo_index: np.ndarray, cell_size, dimension):
    """Order of the final catalogue: by cell index ix*dim0^2 + iy*dim1 + iz, then by
    catalogue index (combine_chunks.py:38-52)."""
    cell_indices = (np.asarray(halo_cofp) // np.asarray(cell_size)).astype("int64")
    assert dimension[0] >= dimension[1] >= dimension[2]
    if cell_indices.shape[0] > 0:
        assert np.min(cell_indices) >= 0
        for i_cell in range(3):
            assert np.max(cell_indices[:, i_cell]) < dimension[i_cell]
    cell_index = (cell_indices[:, 0] * dimension[0] ** 2 + cell_indices[:, 1] * dimension[1]
                  + cell_indices[:, 2])  # fmt: skip
    return np.lexsort((np.asarray(halo_index, dtype=np.int64), cell_index))


def _floor_divide(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """NumPy's `a // b` for doubles (npy_divmod: fmod, then a correction), on the device, so that the cell of a
    halo centre is the one `spatial_sort_order` finds on the host."""
    mod = torch.fmod(a, b)
    div = (a - mod) / b
    adjust = (mod != 0) & ((b < 0) != (mod < 0))
    div = torch.where(adjust, div - 1.0, div)
    floordiv = torch.floor(div)
    floordiv = torch.where(div - floordiv > 0.5, floordiv + 1.0, floordiv)
    return torch.where(div != 0, floordiv, torch.zeros_like(div))


def spatial_sort_order_device(cofp: torch.Tensor, index: torch.Tensor, cell_size, dimension) -> torch.Tensor:
    """`spatial_sort_order` with the halo table resident on the device: two stable sorts (catalogue index,
    then cell index) give the (cell, index) order of combine_chunks.py:38-52."""
    assert dimension[0] >= dimension[1] >= dimension[2]
    cs = torch.as_tensor(np.broadcast_to(np.asarray(cell_size, dtype=np.float64), (3,)).copy(), device=cofp.device)
    ci = _floor_divide(cofp, cs[None, :].expand_as(cofp)).to(torch.int64)
    cell_index = ci[:, 0] * dimension[0] ** 2 + ci[:, 1] * dimension[1] + ci[:, 2]
    o1 = torch.argsort(index, stable=True)
    o2 = torch.argsort(cell_index[o1], stable=True)
    return o1[o2]


def gather_catalogue(table: torch.Tensor, cofp: torch.Tensor, index: torch.Tensor, cell_size,
                     dimension, group=None):  # fmt: skip
    """
    table [n_local, ncols] float64, cofp [n_local, 3] float64, index [n_local] int64, all on
    this rank's device.  Ranks may hold different numbers of halos: rows are padded to the
    maximum for the (single) all-gather and the padding is dropped afterwards.
    Returns (table_all, cofp_all, index_all) in final-catalogue order, identical on all ranks.
    The ordering runs where the table lives: on the GPU for CUDA tensors (nothing but the row
    counts crosses to the host), with NumPy for the gloo / CPU tests.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_local, ncols = table.shape
    dev = table.device

    def order_of(c, i):
        if c.is_cuda:
            return spatial_sort_order_device(c, i, cell_size, dimension)
        return torch.as_tensor(spatial_sort_order(c.numpy(), i.numpy(), cell_size, dimension))

    if world == 1:
        o = order_of(cofp, index)
        return table[o], cofp[o], index[o]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = counts.tolist()
    nmax = max(counts_h)
    # one payload row per halo: [catalogue index bits | cofp(3) | table row], padded to nmax rows
    payload = torch.zeros((nmax, ncols + 4), dtype=torch.float64, device=dev)
    payload[:n_local, 0] = index.to(torch.int64).view(torch.float64) if index.dtype == torch.int64 else index
    payload[:n_local, 1:4] = cofp
    payload[:n_local, 4:] = table
    gathered = torch.empty((world * nmax, ncols + 4), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    if min(counts_h) != nmax:
        keep = torch.cat([torch.arange(int(c), device=dev) + r * nmax for r, c in enumerate(counts_h)])
        gathered = gathered[keep]
    index_all = gathered[:, 0].contiguous().view(torch.int64)
    cofp_all = gathered[:, 1:4].contiguous()
    o = order_of(cofp_all, index_all)
    return gathered[o, 4:], cofp_all[o], index_all[o]


def gather_catalogue_packed(narrow: torch.Tensor, wide: torch.Tensor, cofp: torch.Tensor, index: torch.Tensor,
                             cell_size, dimension, group=None):  # fmt: skip
    """
    `gather_catalogue` for the STORED catalogue (`soap_amd.core.result_set.PackedLayout.pack`): narrow [n_local, Nn]
    float32 and wide [n_local, Nw] float64 instead of the engine's fp64 table.  Two all-gathers (one per dtype; the halo
    keys travel with the wide one), the padding rule and the ordering are those of `gather_catalogue`.
    Returns (narrow_all, wide_all, cofp_all, index_all) in final-catalogue order, identical on all ranks.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_local = narrow.shape[0]
    dev = narrow.device

    def order_of(c, i):
        if c.is_cuda:
            return spatial_sort_order_device(c, i, cell_size, dimension)
        return torch.as_tensor(spatial_sort_order(c.numpy(), i.numpy(), cell_size, dimension))

    if world == 1:
        o = order_of(cofp, index)
        return narrow[o], wide[o], cofp[o], index[o]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = counts.tolist()
    nmax = max(counts_h)
    nw = wide.shape[1]
    pay_w = torch.zeros((nmax, nw + 4), dtype=torch.float64, device=dev)
    pay_w[:n_local, 0] = index.to(torch.int64).view(torch.float64)
    pay_w[:n_local, 1:4] = cofp
    pay_w[:n_local, 4:] = wide
    pay_n = torch.zeros((nmax, narrow.shape[1]), dtype=torch.float32, device=dev)
    pay_n[:n_local] = narrow
    got_w = torch.empty((world * nmax, nw + 4), dtype=torch.float64, device=dev)
    got_n = torch.empty((world * nmax, narrow.shape[1]), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(got_w, pay_w, group=group)
    dist.all_gather_into_tensor(got_n, pay_n, group=group)
    if min(counts_h) != nmax:
        keep = torch.cat([torch.arange(int(c), device=dev) + r * nmax for r, c in enumerate(counts_h)])
        got_w, got_n = got_w[keep], got_n[keep]
    index_all = got_w[:, 0].contiguous().view(torch.int64)
    cofp_all = got_w[:, 1:4].contiguous()
    o = order_of(cofp_all, index_all)
    return got_n[o], got_w[o, 4:], cofp_all[o], index_all[o]
