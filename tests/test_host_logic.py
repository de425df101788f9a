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



def test_bench_parses_the_encoders_ledger_lines_and_matches_profiles_by_source_hash(tmp_path, monkeypatch):
    """bench.py prices its roofline blocks from the X265HIP_VERBOSE lines of the timed encode (x265_amd/host/*.cpp) and quotes PMC traffic only from a
    committed profile whose `# sources` stamp equals the sha256 of the tree's kernel sources."""
    import bench
    lines = [
        "x265hip: lookahead: 2362 frame-cost estimates (556 motion-search passes over 120 lowres frames) served by the GPU in 603 batches, 0.236 s inside the seam",
        "x265hip: lookahead: 504 searches launched ahead of their request, 399 of them used; 30 search launches of 35.3 (frame, reference) pairs on average",
        "x265hip: sadplanes: 7998027 integer-pel SADs of the motion search served from GPU-built SAD surfaces (436 surfaces, 7412 CTU rows in 1322 launches, 88.648 ms of device time), "
        "3836338 of the same searches outside their block's window and 1344 searches without a surface computed on the host",
        "x265hip: device time (HIP events around every launch group): lookahead searches 40.355 ms in 30 launch groups (0 algorithmic bytes), other lookahead kernels 23.703 ms in 603 "
        "launch groups (0 algorithmic bytes), sub-pel plane bands 14.269 ms in 1173 launch groups (2861709312 algorithmic bytes), SAD surfaces 88.648 ms in 1322 launch groups "
        "(110688584400 algorithmic bytes), source energy planes 4.132 ms in 360 launch groups (0 algorithmic bytes); total 171.107 ms",
    ]
    s = bench.parse_served(lines)
    assert s["surfaces"] == 436 and s["ctu_rows"] == 7412 and s["surface_launches"] == 1322
    assert s["search_launches"] == 30 and abs(s["pairs_per_launch"] - 35.3) < 1e-9
    assert abs(s["device_ms"] - 171.107) < 1e-9
    assert s["clocks"]["SAD surfaces"] == {"ms": 88.648, "launch_groups": 1322, "algorithmic_bytes": 110688584400}
    assert s["clocks"]["sub-pel plane bands"]["algorithmic_bytes"] == 2861709312
    # 7412 rows of 30 CTUs at 508 437 bytes, minus the blocks of the ragged last row (1080 = 16 * 64 + 56): the ledger's figure is the library's
    assert 0.95 < s["clocks"]["SAD surfaces"]["algorithmic_bytes"] / (7412 * 30 * 508437.0) <= 1.0

    # profile matching: stamp equal -> quoted, stamp different -> not quoted, and says why
    digest = bench.source_digest("sadsurf.hip")
    prof = tmp_path / "profiles"
    prof.mkdir()
    body = ("# sources sadsurf.hip %s\n# fetch_correction 2.000 write_correction 1.000 (x); per-launch averages in KiB, raw and corrected\n"
            "xh::sadsurf_ctu_kernel   522240   16   8577.6   6588.8   17155.1   6588.8\n")
    (prof / "r99_v1_pmc_sadsurf.txt").write_text(body % digest)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    val, f, note = bench.pmc_profile("r*_pmc_sadsurf.txt", "sadsurf_ctu_kernel", digest)
    assert val == int((17155.1 + 6588.8) * 1024) and f.endswith("r99_v1_pmc_sadsurf.txt") and "fetch_correction" in note
    val, f, note = bench.pmc_profile("r*_pmc_sadsurf.txt", "sadsurf_ctu_kernel", "0" * 16)
    assert val is None and "not quoted" in note


def test_committed_counter_profiles_were_collected_from_this_tree():
    """bench.py quotes `roofline.traffic` only from a committed counter profile whose `# sources` stamp is the hash of the kernel sources in the tree: the
    newest committed profile of each kernel family must carry the current hash — a kernel edit without a re-collection turns the driver's line's traffic to null."""
    import glob
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pattern, kernel, names in (("r*_pmc_sadsurf.txt", "sadsurf_ctu_kernel", ("sadsurf.hip",)),
                                   ("r*_pmc_lookahead.txt", "lookahead_p_kernel", ("lookahead.hip", "lasession.hip")),
                                   ("r*_pmc_encode.txt", "subpel_satd_kernel", ("sadsurf.hip",))):
        val, f, note = bench.pmc_profile(pattern, kernel, bench.source_digest(*names))
        assert val and val > 0, (pattern, f, note)
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_cuserve_pmc_per_job.txt")))
    assert files
    stamp = [l.split()[-1] for l in open(files[-1]) if l.startswith("# sources")]
    assert stamp and stamp[0] == bench.source_digest("cuserve.hip"), (files[-1], stamp)
