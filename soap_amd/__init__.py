"""
soap_amd -- MI355X-native replacement for SOAP's per-halo hot path
(particle_selection + property_calculation, driven as in SOAP/core/halo_tasks.py).

The compute lives in a C-ABI HIP shared library (include/soap_engine.h,
soap_amd/csrc/engine.hip); this package is the Python host side that mirrors the
reference's operator surface (HaloProperty.calculate / process_halos).
There is NO CPU fallback: importing works anywhere, computing needs the built
library and a GPU and fails loudly otherwise.
"""

__version__ = "0.1.0"
