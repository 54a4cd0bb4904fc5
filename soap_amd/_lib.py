"""
ctypes binding of the C ABI in include/soap_engine.h.

The shared library is built in-tree (soap_amd/lib/libsoap_engine.so) by
`__graft_entry__.build()` / `make -C soap_amd/csrc`.  There is no fallback: if it
is missing, `load()` raises.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsoap_engine.so")
ABI_VERSION = 5
NTYPES = 4


class EngineConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device", C.c_int32),
        ("boxsize", C.c_double),
        ("G_eff", C.c_double),
        ("nu_density", C.c_double),
        ("mpc20", C.c_double),
        ("a", C.c_double),
        ("mpc", C.c_double),
        ("softening", C.c_double * NTYPES),
        ("n_workgroups", C.c_int32),
        ("slab_capacity", C.c_int32),
        ("G_phys", C.c_double),
        ("hubble", C.c_double),
        ("rdot_omega", C.c_double),
    ]


class CalcDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("parent", C.c_int32),
        ("reference_density", C.c_double),
        ("radius", C.c_double),
        ("physical_radius_mpc", C.c_double),
        ("prev_radius", C.c_double),
        ("virial_definition", C.c_int32),
        ("filter_limit", C.c_int32),
        ("filter_types", C.c_int32),
        ("flags", C.c_int32),
    ]


# every symbol include/soap_engine.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "soap_engine_abi_version": (C.c_int32, []),
    "soap_engine_last_error": (C.c_char_p, []),
    "soap_engine_calc_columns": (C.c_int32, [C.c_int32]),
    "soap_engine_desc_columns": (C.c_int32, [C.POINTER(CalcDesc)]),
    "soap_engine_create": (C.c_int32, [C.POINTER(EngineConfig), C.POINTER(C.c_void_p)]),
    "soap_engine_destroy": (C.c_int32, [C.c_void_p]),
    "soap_engine_stage_particles": (
        C.c_int32,
        [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
         C.c_int32, C.c_void_p, C.c_int32, C.c_int32],
    ),
    "soap_engine_wait_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "soap_engine_stage_neutrinos": (
        C.c_int32,
        [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32],
    ),  # fmt: skip
    "soap_engine_stage_halos": (
        C.c_int32,
        [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),  # fmt: skip
    "soap_engine_set_resident_radius": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "soap_engine_set_member_probe": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64]),
    "soap_engine_fetch_members": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "soap_engine_run": (C.c_int32, [C.c_void_p, C.POINTER(CalcDesc), C.c_int32, C.c_double]),
    "soap_engine_row_columns": (C.c_int64, [C.c_void_p]),
    "soap_engine_fetch": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "soap_engine_table_device": (C.c_void_p, [C.c_void_p]),
    "soap_engine_copy_table_device": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "soap_engine_pack_catalogue": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
         C.c_void_p],
    ),  # fmt: skip
    "soap_engine_membership": (
        C.c_int64,
        [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_void_p,
         C.c_int64],
    ),  # fmt: skip
    "soap_engine_kernel_ms": (C.c_double, [C.c_void_p, C.c_char_p]),
    "soap_engine_phase_cycles": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

_lib = None


class EngineLibraryMissing(RuntimeError):
    pass


def load():
    """Load the HIP engine library; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SOAP_ENGINE_LIB", LIB_PATH)   # (another build of the same sources, e.g. scripts/build_variant.sh)
    if not os.path.exists(path):
        raise EngineLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (there is no CPU fallback)"
        )
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.soap_engine_abi_version() != ABI_VERSION:
        raise EngineLibraryMissing("libsoap_engine.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def last_error() -> str:
    return load().soap_engine_last_error().decode()
