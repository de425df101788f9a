"""CPU tests of the product's host-side logic (no GPU needed): BitCost::setQP restatement, frame-pass geometry."""
import numpy as np
import pytest

from frame_oracle import counts, make_scene, oracle_frame_pass
from oracle import pyoracle as po


def test_product_mvcost_table_equals_pinned_oracle_for_every_qp():
    from x265_amd import hipprim as hp
    L = hp.lib()
    t = np.zeros(4 * 32768 + 1, np.uint16)
    for depth in (8, 10, 12):
        for qp in range(0, 52):
            assert L.x265hip_mvcost_table(qp, depth, t.ctypes.data, 2 * 32768) == 0
            assert np.array_equal(t, po.mvcost_table(qp, depth)), (depth, qp)
    assert L.x265hip_mvcost_table(99, 8, t.ctypes.data, 2 * 32768) == -1


def test_algorithmic_bytes_accounting():
    from x265_amd.framepass import algorithmic_bytes
    b = algorithmic_bytes(1920, 1080, 8)
    ncu, ntu = counts(1920, 1080)
    assert ncu == [480, 1980, 8040, 32400] and ntu == [1980, 720]
    assert b["chain"] == (1980 * 1024 + 720 * 64) * 5
    assert b["sa8d"] == sum(n * (2 * s * s + 4) for n, s in zip(ncu, (64, 32, 16, 8)))


def test_oracle_frame_pass_properties():
    """Sanity of the checker itself: deterministic, recon margins are edge replicas, static scene -> zero vectors."""
    sc = make_scene(200, 136, depth=8, seed=3, tile=48)
    a = oracle_frame_pass(sc["src"], sc["ref"])
    b = oracle_frame_pass(sc["src"], sc["ref"])
    assert all(np.array_equal(x, y) for x, y in zip(a["mv"], b["mv"]))
    m = 96
    core = a["recon"][m:-m, m:-m]
    assert np.array_equal(a["recon"], np.pad(core, m, mode="edge"))
    still = oracle_frame_pass(sc["ref"], sc["ref"])
    assert all(not x.any() for x in still["mv"]) and all(not x.any() for x in still["numSig"])
    assert np.array_equal(still["recon"][m:-m, m:-m], sc["ref"])


def test_bench_parallel_cpu_baseline_leg_runs_and_is_bounded():
    """bench.py's multi-process CPU leg (plain subprocesses with a hard timeout): two workers, one frame pass each, finishes and reports."""
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    t0 = time.time()
    r = bench.cpu_baseline_parallel(frames_per_chain=1, max_procs=2, timeout_s=90, start_delay_s=6.0)
    assert "error" not in r, r
    assert r["cores"] == 2 and r["value"] > 0 and r["kind"] == "port"
    assert time.time() - t0 < 90

