"""
Thin object wrapper over the C ABI (include/soap_engine.h).  Accepts NumPy arrays (host
pointers) or torch CUDA tensors (device pointers; torch is only plumbing for HBM
residency, it never computes anything on this path).
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import numpy as np

from . import _lib
from ._lib import CalcDesc, EngineConfig
from .layout import STATS_COLUMNS

STATUS_OK = 0
STATUS_NEED_LARGER_READ_RADIUS = 1
STATUS_FATAL_TOO_MANY_BOUND = -1
STATUS_FATAL_SO_BEYOND_20MPC = -2
STATUS_FATAL_SLAB_OVERFLOW = -3
STATUS_FATAL_RADIUS_ASSERT = -4

# type slot <-> SWIFT particle type
PTYPE_OF_SLOT = (0, 1, 4, 5)
SLOT_OF_PTYPE = {0: 0, 1: 1, 4: 2, 5: 3}


# slots of soap_engine_phase_cycles; entries from index 8 on are sub-phases (nested inside the first eight)
PHASE_NAMES = ("select", "sort", "materialize", "reduce", "scan", "so_search", "total", "misc",
               "reduce_xlane", "finalize_moments", "finalize_scan", "scan_argmax", "mat_cumsum", "so_thread0",
               "family_order", "stage_a", "prologue", "epilogue", "subhalo_spin", "projected",
               "stage_b", "scan_setup", "sort_histogram", "sort_scatter", "sort_blocks", "flow_inner", "flow_shells", "rotation") + ("",) * 3 + ("exact_cumsum_chunks",)  # fmt: skip
# (the last slot is a count, not clocks: chunks of the slab build whose cumulative mass was summed in parallel because every partial sum is exact)


class EngineError(RuntimeError):
    pass


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _ptr(x, dtype, shape_tail=()):
    """(pointer, on_device, keepalive) of a contiguous array of `dtype`."""
    if _is_torch(x):
        import torch

        tdt = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32}[dtype]
        if x.dtype != tdt:
            raise TypeError(f"expected torch dtype {tdt}, got {x.dtype}")
        if not x.is_contiguous():
            x = x.contiguous()
        return C.c_void_p(x.data_ptr()), bool(x.is_cuda), x
    a = np.ascontiguousarray(x, dtype=dtype)
    return C.c_void_p(a.ctypes.data), False, a


def _value_dtype(*arrays):
    """Width the value arrays of one staging call go through the C ABI with (`value_bytes` of
    soap_engine_stage_particles): float32 when every one of them is float32 -- what SWIFT snapshots hold for
    Velocities and Masses; the engine widens on the device -- float64 otherwise (a float32 array among float64
    ones is widened here)."""
    def is32(x):
        if _is_torch(x):
            import torch

            return x.dtype == torch.float32
        return np.asarray(x).dtype == np.float32

    return np.float32 if all(is32(x) for x in arrays) else np.float64


def _as_value(x, dtype):
    if _is_torch(x):
        import torch

        return x if dtype == np.float32 else x.to(torch.float64)
    return x


def mesh_resolution(nr_parts: int) -> int:
    """Cells per dimension for ~1000 particles per cell, at most 256
    (SOAP/core/chunk_tasks.py:299-302)."""
    resolution = int((nr_parts / 1000) ** (1.0 / 3.0))
    return min(max(resolution, 1), 256)


class Engine:
    """One GPU's halo-property engine (one handle = one chunk at a time)."""

    def __init__(self, boxsize, G_eff, softening, nu_density=0.0, mpc20=float("inf"),
                 a=1.0, mpc=1.0, device=0, n_workgroups=0, slab_capacity=0,
                 G_phys=0.0, hubble=0.0, rdot_omega=0.0):  # fmt: skip
        self._lib = _lib.load()
        cfg = EngineConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.device = int(device)
        cfg.boxsize = float(boxsize)
        cfg.G_eff = float(G_eff)
        cfg.nu_density = float(nu_density)
        cfg.mpc20 = float(mpc20)
        cfg.a = float(a)
        cfg.mpc = float(mpc)
        for i in range(_lib.NTYPES):
            cfg.softening[i] = float(softening[i])
        cfg.n_workgroups = int(n_workgroups)
        cfg.slab_capacity = int(slab_capacity)
        cfg.G_phys = float(G_phys)
        cfg.hubble = float(hubble)
        cfg.rdot_omega = float(rdot_omega)
        self._h = C.c_void_p()
        self._in_flight = None   # arrays of a staging call whose placement pass is still running on the device
        self._check(self._lib.soap_engine_create(C.byref(cfg), C.byref(self._h)))
        self.n_halo = 0
        self.calcs = []

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"soap_engine error {rc}: {_lib.last_error()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.soap_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging ---------------------------------------------------------------------
    def stage_particles(self, type_slot, pos, vel, mass, grnr, fof, resolution=None, asynchronous=True):
        """`asynchronous` (device arrays only): return with the placement pass still in flight (on_device = 2 of
        soap_engine_stage_particles); the engine keeps references to the arrays until the next `run` has completed the call, the
        caller must not write into them before that."""
        n = int(pos.shape[0])
        if resolution is None:
            resolution = mesh_resolution(n)
        vdt = _value_dtype(vel, mass)
        p_pos, dev, k1 = _ptr(pos, np.float64)
        p_vel, dev2, k2 = _ptr(_as_value(vel, vdt), vdt)
        p_m, dev3, k3 = _ptr(_as_value(mass, vdt), vdt)
        if _is_torch(grnr):
            import torch

            gb = 8 if grnr.dtype == torch.int64 else 4
        else:
            grnr = np.asarray(grnr)
            gb = 8 if grnr.dtype == np.int64 else 4
        p_g, dev4, k4 = _ptr(grnr, np.int64 if gb == 8 else np.int32)
        p_f, dev5, k5 = _ptr(fof, np.int64)
        if len({dev, dev2, dev3, dev4, dev5}) != 1:
            raise ValueError("all particle arrays must live on the same side (host or device)")
        if tuple(pos.shape) != (n, 3) or tuple(vel.shape) != (n, 3):
            raise ValueError("Coordinates/Velocities must have shape (n, 3)")
        if dev:
            self._order_after_torch(pos)
        mode = (2 if asynchronous and n > 0 and os.environ.get("SOAP_ENGINE_ASYNC_STAGE", "1") != "0" else 1) if dev else 0
        self._check(
            self._lib.soap_engine_stage_particles(
                self._h, int(type_slot), n, p_pos, p_vel, p_m, 4 if vdt == np.float32 else 8, p_g, gb, p_f,
                int(resolution), mode,
            )  # fmt: skip
        )
        if mode == 2:
            # still being read on the engine's stream: the arrays (and their widened / contiguous copies) live until the next
            # call that completes the staging (include/soap_engine.h, lifetime rule)
            self._in_flight = [pos, vel, mass, grnr, fof, k1, k2, k3, k4, k5]
        del k1, k2, k3, k4, k5

    def _order_after_torch(self, tensor):
        """Device inputs are read on the engine's own stream: order it behind whatever torch has submitted to its
        current stream (the producer of the tensors, their `.contiguous()` / widening copies) -- an event on the
        device, not a host synchronisation per array."""
        import torch

        self._check(self._lib.soap_engine_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream)))

    def clear_type(self, type_slot):
        """Mark a particle type as absent from the current chunk (type_slot 0..3, or "nu")."""
        if type_slot == "nu":
            self._check(self._lib.soap_engine_stage_neutrinos(self._h, 0, None, None, None, 8, 1, 0))
        else:
            self._check(self._lib.soap_engine_stage_particles(self._h, int(type_slot), 0, None, None, None, 8, None, 4, None, 1, 0))

    def stage_neutrinos(self, pos, mass, weight, resolution=None):
        """PartType6: Coordinates, Masses, Weights (SO mass profile only, SO_properties.py:380-395)."""
        n = int(pos.shape[0])
        if resolution is None:
            resolution = mesh_resolution(n)
        vdt = _value_dtype(mass, weight)
        p_pos, dev, k1 = _ptr(pos, np.float64)
        p_m, dev2, k2 = _ptr(_as_value(mass, vdt), vdt)
        p_w, dev3, k3 = _ptr(_as_value(weight, vdt), vdt)
        if len({dev, dev2, dev3}) != 1:
            raise ValueError("all particle arrays must live on the same side (host or device)")
        if n and tuple(pos.shape) != (n, 3):
            raise ValueError("Coordinates must have shape (n, 3)")
        if dev:
            self._order_after_torch(pos)
        self._check(self._lib.soap_engine_stage_neutrinos(self._h, n, p_pos, p_m, p_w, 4 if vdt == np.float32 else 8, int(resolution), 1 if dev else 0))
        del k1, k2, k3

    def stage_halos(self, cofp, search_radius, read_radius, index, is_central, nr_bound_part, resident_radius=None):
        """`resident_radius` (optional, per halo, >= read_radius): how far around the centre the staged particles
        are complete; lets the engine apply read-radius updates itself (soap_engine_set_resident_radius)."""
        n = int(np.asarray(index).shape[0])
        a = [
            np.ascontiguousarray(cofp, dtype=np.float64).reshape(n, 3),
            np.ascontiguousarray(search_radius, dtype=np.float64),
            np.ascontiguousarray(read_radius, dtype=np.float64),
            np.ascontiguousarray(index, dtype=np.int64),
            np.ascontiguousarray(is_central, dtype=np.int32),
            np.ascontiguousarray(nr_bound_part, dtype=np.int64),
        ]
        self._check(
            self._lib.soap_engine_stage_halos(self._h, n, *[C.c_void_p(x.ctypes.data) for x in a])
        )
        self.n_halo = n
        if resident_radius is not None and n > 0:
            rr = np.ascontiguousarray(np.broadcast_to(np.asarray(resident_radius, dtype=np.float64), (n,)))
            if np.any(rr < a[2]):
                raise ValueError("resident_radius must be >= read_radius")
            self._check(self._lib.soap_engine_set_resident_radius(self._h, C.c_void_p(rr.ctypes.data)))

    # -- verification probe ----------------------------------------------------------------
    def set_member_probe(self, row, calc, cap=1 << 22):
        """Record, during the next run, which staged particles calculation number `calc` of halo row `row` sums over
        (row=None switches the probe off)."""
        self._check(self._lib.soap_engine_set_member_probe(self._h, -1 if row is None else int(row), int(calc), int(cap)))

    def fetch_members(self, cap=1 << 22):
        """(type_slot, index in the staged arrays of that type) of the probed calculation's members, sorted."""
        ts = np.zeros(cap, dtype=np.int32)
        ix = np.zeros(cap, dtype=np.int64)
        n = self._lib.soap_engine_fetch_members(self._h, C.c_void_p(ts.ctypes.data), C.c_void_p(ix.ctypes.data), cap)
        if n < 0:
            raise EngineError(f"fetch_members failed: {_lib.last_error()}")
        if n > cap:
            raise EngineError(f"member probe capacity {cap} too small for {n} members")
        return ts[:n].copy(), ix[:n].copy()

    # -- run / fetch -------------------------------------------------------------------
    MAX_CALCULATIONS = 64   # MAXCALC of the kernels: one lane of a wavefront and one bit of the done mask per calculation

    def run(self, calcs: Sequence[CalcDesc], target_density):
        if len(calcs) > self.MAX_CALCULATIONS:
            raise EngineError(
                f"{len(calcs)} calculations in halo_prop_list, the engine takes at most {self.MAX_CALCULATIONS} per run: "
                "split the list into several runs, each starting with the bound subhalo (INTEGRATION.md)")
        arr = (CalcDesc * len(calcs))(*calcs)
        td = -1.0 if target_density is None else float(target_density)
        try:
            self._check(self._lib.soap_engine_run(self._h, arr, len(calcs), td))
        finally:
            self._in_flight = None   # (the run has completed a staging call that was still in flight)
        self.calcs = list(calcs)

    def desc_columns(self, desc) -> int:
        """Columns the calculation `desc` (a CalcDesc) occupies in the result table of a run."""
        return int(self._lib.soap_engine_desc_columns(C.byref(desc)))

    @property
    def row_columns(self):
        return int(self._lib.soap_engine_row_columns(self._h))

    def fetch(self, table=True):
        """Copy status/radii/stats (and, unless table=False, the fp64 result table) to the host."""
        n, nc = self.n_halo, self.row_columns
        if not table:
            return self._fetch_small()
        table = np.zeros((n, nc), dtype=np.float64)
        status = np.zeros(n, dtype=np.int32)
        search = np.zeros(n, dtype=np.float64)
        read = np.zeros(n, dtype=np.float64)
        stats = np.zeros((n, STATS_COLUMNS), dtype=np.int64)
        if n > 0:
            self._check(
                self._lib.soap_engine_fetch(
                    self._h, *[C.c_void_p(x.ctypes.data) for x in (table, status, search, read, stats)]
                )
            )
        return dict(table=table, status=status, search_radius=search, read_radius=read, stats=stats)

    def _fetch_small(self):
        n = self.n_halo
        status = np.zeros(n, dtype=np.int32)
        search = np.zeros(n, dtype=np.float64)
        read = np.zeros(n, dtype=np.float64)
        stats = np.zeros((n, STATS_COLUMNS), dtype=np.int64)
        if n > 0:
            self._check(
                self._lib.soap_engine_fetch(
                    self._h, None, *[C.c_void_p(x.ctypes.data) for x in (status, search, read, stats)]
                )
            )
        return dict(table=None, status=status, search_radius=search, read_radius=read, stats=stats)

    def copy_table_to(self, tensor):
        """Device-to-device copy of the result table into a torch CUDA float64 tensor."""
        if tensor.numel() != self.n_halo * self.row_columns or not tensor.is_cuda:
            raise ValueError("need a CUDA float64 tensor of n_halo*row_columns elements")
        self._check(self._lib.soap_engine_copy_table_device(self._h, C.c_void_p(tensor.data_ptr())))

    def pack_catalogue(self, rows, narrow_cols, narrow_scale, wide_cols, wide_scale, out_narrow, out_wide):
        """Stored catalogue of the last run from the engine's table in ONE pass (soap_engine_pack_catalogue): torch CUDA
        tensors -- rows int64 [n] or None, *_cols int32, *_scale float64, out_narrow float32 [n, Nn], out_wide float64 [n, Nw]."""
        n = int(out_narrow.shape[0])
        self._order_after_torch(out_narrow)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None  # noqa: E731
        self._check(self._lib.soap_engine_pack_catalogue(
            self._h, ptr(rows), n, ptr(narrow_cols), ptr(narrow_scale), int(narrow_cols.numel()), ptr(out_narrow),
            ptr(wide_cols), ptr(wide_scale), int(wide_cols.numel()), ptr(out_wide)))

    def table_device_ptr(self):
        return self._lib.soap_engine_table_device(self._h)

    def membership(self, row, type_slot, radius, strict, bound_only=False, cap=1 << 22):
        out = np.zeros(cap, dtype=np.int64)
        n = self._lib.soap_engine_membership(
            self._h, int(row), int(type_slot), float(radius), 1 if strict else 0,
            1 if bound_only else 0, C.c_void_p(out.ctypes.data), cap,
        )  # fmt: skip
        if n < 0:
            raise EngineError(f"membership failed: {_lib.last_error()}")
        if n > cap:
            return self.membership(row, type_slot, radius, strict, bound_only, cap=int(n))
        return out[:n].copy()

    def phase_cycles(self):
        out = (C.c_uint64 * 32)()
        self._check(self._lib.soap_engine_phase_cycles(self._h, out))
        return {n: int(v) for n, v in zip(PHASE_NAMES, out) if n}

    def kernel_ms(self, name: str) -> float:
        return float(self._lib.soap_engine_kernel_ms(self._h, name.encode()))
