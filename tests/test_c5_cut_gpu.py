"""
BASELINE config 5 AS A CUT OF THE 2e8 BOX: the synthetic hydro box of configs[4] (2e8 particles of four types -- gas, dark matter,
stars, black holes --, 2e5 halos, masses float32-valued as SWIFT stores them; BoundSubhalo + 9 SO + inclusive / exclusive 30 / 50 /
100 kpc + projected 10 / 30 / 50 / 100 kpc) is generated whole on the one GPU of the test box, run unsplit, and then ranks 0 and 7 of its
8-way Peano-Hilbert cut (SOAP/core/domain_decomposition.py:64-142 through soap_amd.core.chunk_tasks.split_halos) are run as a rank
of the 8-GPU job runs them: cell-masked read (mask_cells.py:6-38), coordinates wrapped about the chunk (chunk_tasks.py:48-50),
resident radii from the mask, the chunk loop with read-radius retries.  Every row of both ranks against the same halo's row of
the unsplit run: statuses, radius rounds, sphere contents, particle counts and radii exactly, values to 1e-10 (the comparison of
tests/test_c4_at_size_gpu.py).  `bench.WORKLOADS["c5"]` is an independently generated 2.5e7-particle box of the same recipe -- the
one-GPU share the bench line times; this test is the cut itself.  What is left to the hardware run is the RCCL gather.
"""

import pytest

from soap_amd.synthetic import make_chunk
from tests.test_c4_at_size_gpu import cut_ranks_equal_the_unsplit_box

pytestmark = pytest.mark.gpu


def test_ranks_0_and_7_of_the_8_way_cut_of_the_2e8_hydro_box_equal_the_unsplit_run(built_library):
    import bench

    wl = dict(bench.WORKLOADS["c5"], n_part=200_000_000, n_halo=200_000)
    sc, data, halos = make_chunk(wl["n_part"], wl["n_halo"], seed=1234, device="cuda:0", hydro=True, topcells=bench.TOPCELLS,
                                 float32_masses=True)
    props = bench.build_props(sc, wl)[4]
    cut_ranks_equal_the_unsplit_box(sc, data, halos, props, (0, 7), "c5 cut")
