"""Time the mesh kernels alone with an alternative library (diagnostic): LIB=path python scripts/mesh_probe.py"""
import os, sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
from soap_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
from soap_amd.core import halo_tasks
from soap_amd.synthetic import make_chunk
import bench
wl = bench.WORKLOADS["c3"]
sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0")
eng = halo_tasks.make_engine(sc, device=0)
for rep in range(3):
    halo_tasks.stage_chunk(eng, data)
    print("mesh", round(eng.kernel_ms("mesh"), 2), "reorder", round(eng.kernel_ms("reorder"), 2), flush=True)
