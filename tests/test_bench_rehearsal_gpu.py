"""
The N > 1 path of bench.py as the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`),
rehearsed with TWO ranks on the one GPU of the test box: BENCH_REHEARSAL=1 puts both ranks on cuda:0 and uses gloo for the
barrier, the max-over-ranks clock and the catalogue gather -- everything else (Peano-Hilbert cut of the halo table, cell-masked
read of each rank's region, resident radii, per-rank engine, gathered catalogue) is the code an 8-GPU node runs over RCCL.
A rehearsal, not a scaling measurement: both ranks share one GPU.
"""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]   # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_rank_rehearsal_prints_the_line_of_the_whole_job(built_library):
    env = dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", BENCH_RANK_CUT="work")
    # the configuration of the driver's scaling run: the seed-1234 C3 box (1e8 particles, 1e5 halos), both ranks on this one GPU
    common = ["--steps", "1", "--warmup", "0", "--passes-per-step", "1", "--workload", "c3", "--no-cpu-baseline"]
    port = 29700 + (os.getpid() % 2000)
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "2"] + common, env)  # fmt: skip
    one = _line([sys.executable, "bench.py", "--gpus", "1", "--no-full-set"] + common, env)
    for ln, n in ((one, 1), (two, 2)):
        assert ln["n_gpus"] == n and ln["config"]["chunks"] == n and ln["halos_unfinished"] == 0
        assert ln["metric"] == "particle_aperture_evals_per_s" and ln["scaling"] == "strong" and ln["steps"] == 1
    assert two["config"]["rank_cut"] == "work" and one["config"]["rank_cut"] is None
    # every halo exactly once: the job's evaluations per pass (value x seconds per pass) are those of the unsplit run
    ev1 = one["value"] * one["ms_per_pass"] * 1e-3
    ev2 = two["value"] * two["ms_per_pass"] * 1e-3
    assert abs(ev1 - ev2) <= 1e-9 * ev1, (ev1, ev2)
    assert abs(one["halos_per_s"] * one["ms_per_pass"] - two["halos_per_s"] * two["ms_per_pass"]) <= 1e-6 * one["halos_per_s"] * one["ms_per_pass"]
    # boundary cells are held by both ranks, the rest of the box by one
    assert one["replication"] == 1.0 and 1.0 < two["replication"] < 1.7, two["replication"]
