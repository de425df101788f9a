"""Pin oracle/x265_oracle.c to the REAL reference (oracle/_ref/libx265ref{8,10}.so, built from
/root/reference/source by oracle/Makefile): every primitive, every PU/CU/TU size, TestBench input distributions,
bit-exact.  Skipped only where oracle/_ref was never built (it travels to the GPU box as a prebuilt .so)."""
import numpy as np
import pytest

from backends import Orc, Ref, same
from cases import gen_cases, me_scene
from oracle import pyoracle as po

DEPTHS = [8, 10, 12]          # Main, Main10, Main12 builds of the reference (oracle/Makefile ref8 / ref10 / ref12)


def _need_ref(depth):
    if not po.ref_available(depth):
        pytest.skip("oracle/_ref/libx265ref%d.so not built (make -C oracle ref)" % depth)


@pytest.mark.parametrize("depth", DEPTHS)
def test_tables_match_reference(depth):
    _need_ref(depth)
    o, r = po.oracle(), po.ref(depth)
    for log2n in (2, 3, 4, 5):
        n = 1 << (2 * log2n)
        assert [o.orc_dct_matrix(log2n)[i] for i in range(n)] == [r.ref_dct_matrix(log2n)[i] for i in range(n)]
    for i in range(4):
        assert [o.orc_luma_filter(i)[k] for k in range(8)] == [r.ref_luma_filter(i)[k] for k in range(8)]
    for i in range(8):
        assert [o.orc_chroma_filter(i)[k] for k in range(4)] == [r.ref_chroma_filter(i)[k] for k in range(4)]
    for (w, h) in po.PU_SIZES:
        assert o.orc_partition_from_sizes(w, h) == r.ref_partition_from_sizes(w, h)


@pytest.mark.parametrize("depth", DEPTHS)
def test_mvcost_table_matches_bitcost(depth):
    _need_ref(depth)
    for qp in (0, 12, 22, 28, 37, 51):
        assert np.array_equal(Orc(depth).mvcost_table(qp), Ref(depth).mvcost_table(qp)), qp


@pytest.mark.parametrize("depth", DEPTHS)
def test_every_primitive_matches_reference(depth):
    _need_ref(depth)
    o, r = Orc(depth), Ref(depth)
    n = 0
    for label, fn, args in gen_cases(depth):
        a, b = getattr(o, fn)(*args), getattr(r, fn)(*args)
        assert same(a, b), "%s (depth %d): oracle != reference" % (label, depth)
        n += 1
    assert n > 2000


@pytest.mark.parametrize("depth", DEPTHS)
def test_loop_filter_primitives_match_reference(depth):
    """pelFilterLumaStrong / pelFilterChroma, calSign, saoCuOrgE0..E3 / B0 and saoCuStatsBO / E0..E3 of the reference's C table vs the restatement."""
    _need_ref(depth)
    from cases import loop_cases, deblock_cases
    import itertools
    o, r = Orc(depth), Ref(depth)
    n = 0
    # deblock_cases: the real Deblock::edgeFilterLuma / edgeFilterChroma on a one-CTU CUData vs the per-unit restatement
    for label, fn, args in itertools.chain(loop_cases(depth), deblock_cases(depth)):
        assert same(getattr(o, fn)(*args), getattr(r, fn)(*args)), label
        n += 1
    assert n >= 380


@pytest.mark.parametrize("depth", DEPTHS)
def test_weightp_analysis_matches_reference(depth):
    """The real LookaheadTLD::weightsAnalyse (real Lowres objects, the lookahead's own weightCostLuma) vs the restatement: the decision and,
    where it weights, the four weighted lowres planes; fades that must be weighted and pairs that must not."""
    _need_ref(depth)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.weightp_results(Orc, depth), make_golden.weightp_results(Ref, depth)
    assert sum(v[0] for v in b.values()) >= 4 and sum(1 - v[0] for v in b.values()) >= 2
    for k in a:
        assert same(a[k], b[k]), k
    # and the whole P-frame estimate with --weightp: the reference's estimateFrameCost (weightsAnalyse inside, list 0 on the weighted planes)
    a, b = make_golden.lookahead_weightp_results(Orc, depth), make_golden.lookahead_weightp_results(Ref, depth)
    for k in a:
        assert all(same(x, y) for x, y in zip(a[k], b[k])) and len(a[k]) == len(b[k]), k
    assert sum(v[-1] for v in b.values()) >= 4
    # adaptive quantisation: the real calcAdaptiveQuantFrame (a Frame around the test planes) — doubles, ints and the frame statistics
    a, b = make_golden.aq_results(Orc, depth), make_golden.aq_results(Ref, depth)
    for k in a:
        assert a[k][0] == b[k][0] and all(np.array_equal(x, y) for x, y in zip(a[k][1:], b[k][1:])), k


def test_cutree_matches_reference():
    """The real Lookahead::estimateCUPropagate and cuTreeFinish on real Lowres objects vs the restatement: the references' propagateCost after
    the step (incl. saturation and vectors leaving the frame) and the frame's qpCuTreeOffset."""
    _need_ref(8)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.cutree_results(Orc), make_golden.cutree_results(Ref)
    for k in a:
        assert all(np.array_equal(x, y) for x, y in zip(a[k], b[k])), k


def test_coefficient_scan_primitives_match_reference():
    """scanPosLast / findPosFirstLast / costCoeffNxN / costCoeffRemain / costC1C2Flag of the reference's C table and its scan-order tables vs the
    restatement, on inputs drawn like test/pixelharness.cpp draws them; and the committed CABAC cost table is the reference's."""
    _need_ref(8)
    from cases import coef_cases
    from backends import entropy_state_bits_fixture, scan_order_py
    o, r = Orc(8), Ref(8)
    assert np.array_equal(entropy_state_bits_fixture(), r.entropy_state_bits())
    for t in range(3):
        assert np.array_equal(o.scan4x4(t), r.scan4x4(t))
        for log2 in (2, 3, 4, 5):
            assert np.array_equal(o.scan_order(t, log2), r.scan_order(t, log2)) and np.array_equal(scan_order_py(t, log2), r.scan_order(t, log2))
    n = 0
    for label, fn, args in coef_cases():
        assert same(getattr(o, fn)(*args), getattr(r, fn)(*args)), label
        n += 1
    assert n >= 600


@pytest.mark.parametrize("depth", DEPTHS)
def test_motion_compensation_matches_reference(depth):
    """The real Predict::motionCompensation (one-PU CUData / Slice / PPS around the test planes) vs the restatement, every branch."""
    _need_ref(depth)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.mc_results(Orc, depth), make_golden.mc_results(Ref, depth)
    assert len(a) >= 240
    bad = [k for k in a if not same(a[k], b[k])]
    assert not bad, bad[:8]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [1, 2, 3])
def test_search_methods_random_stress_matches_reference(depth, method):
    """HEX / UMH / STAR on smooth scenes with noise from none to heavy (so that the SAD thresholds of UMH and the early exits of STAR go both
    ways), small and large ranges, tight vertical bounds, predictors near and far from the true motion: 240 random PUs per case."""
    _need_ref(depth)
    o, r = Orc(depth), Ref(depth)
    sc = 1 << (depth - 8)
    pmax = (1 << depth) - 1
    n = 0
    for seed in range(6):
        rng = np.random.default_rng(7000 + 10 * seed + method)
        sigma = [0, 0.5, 1, 3, 8, 30][seed] * sc
        shift = (int(rng.integers(-9, 10)), int(rng.integers(-9, 10)))
        m, H, W = 96, 192, 192
        base = rng.integers(0, 256, size=((H + 2 * m) // 8 + 2, (W + 2 * m) // 8 + 2)).astype(np.float64)
        big = np.kron(base, np.ones((8, 8)))[:H + 2 * m, :W + 2 * m]
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, 3, 0) + np.roll(big, 3, 1)) / 5
        refp = np.clip(np.rint((big + rng.normal(0, 2, big.shape)) * sc), 0, pmax).astype(o.pix)
        srcp = np.clip(np.rint(np.roll(refp.astype(np.float64), shift, (0, 1)) + rng.normal(0, sigma, big.shape)), 0, pmax).astype(o.pix)
        for _ in range(40):
            w, h = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 24), (64, 32), (24, 32), (4, 8), (8, 4), (16, 12)][int(rng.integers(0, 12))]
            bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
            by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
            merange = int(rng.choice([4, 8, 16, 32, 57]))
            if rng.integers(0, 2):
                qmvp = (int(-shift[1] * 4 + rng.integers(-6, 7)), int(-shift[0] * 4 + rng.integers(-6, 7)))
            else:
                qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            mvmin = [(qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange]
            mvmax = [(qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange]
            k = int(rng.integers(0, 4))
            if k == 0:
                mvmax[1] = max(min(mvmax[1], int(rng.integers(0, 6))), mvmin[1])
            if k == 1:
                mvmin[1] = min(max(mvmin[1], int(rng.integers(-3, 4))), mvmax[1])
            mvc = [((int(-shift[1] * 4 + rng.integers(-8, 9)), int(-shift[0] * 4 + rng.integers(-8, 9))) if rng.integers(0, 2)
                    else (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))) for _ in range(int(rng.integers(0, 5)))]
            subme, qp = int(rng.choice([0, 2, 3, 7])), int(rng.choice([22, 28, 37]))
            a = o.motion_estimate(refp, srcp, bx, by, w, h, tuple(mvmin), tuple(mvmax), qmvp, mvc, merange, method, subme, qp)
            b = r.motion_estimate(refp, srcp, bx, by, w, h, tuple(mvmin), tuple(mvmax), qmvp, mvc, merange, method, subme, qp)
            assert a == b, (depth, method, seed, w, h, bx, by, merange, qmvp, mvmin, mvmax, mvc, subme, a, b)
            n += 1
    assert n == 240


@pytest.mark.parametrize("depth", DEPTHS)
def test_umh_search_matches_reference(depth):
    """X265_UMH_SEARCH on scenes built to reach its early-termination, cross and adaptive-range branches (tests/cases.py umh_scenes)."""
    _need_ref(depth)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.umh_results(Orc, depth), make_golden.umh_results(Ref, depth)
    assert len(a) >= 300
    assert a == b, [k for k in a if a[k] != b[k]][:8]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [0, 1, 2, 3, 5])     # DIA, HEX, UMH, STAR, FULL (x265.h X265_*_SEARCH)
def test_motion_estimate_matches_reference(depth, method):
    _need_ref(depth)
    o, r = Orc(depth), Ref(depth)
    rng = np.random.default_rng(77 + depth + method)
    refp, srcp, m = me_scene(depth, 99 + depth)
    H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 24), (12, 16), (64, 48), (16, 4), (8, 32)]
    n = 0
    for subme in (0, 1, 2, 3, 5, 7):
        for (w, h) in sizes:
            for _ in range(2 if method != 5 else 1):
                bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                merange = 12 if method == 5 else int(rng.choice([16, 32, 57]))
                qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                # search window as Search::setSearchRange does it (search.cpp:2724): mvp +- merange, full-pel
                mvmin = ((qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange)
                mvmax = ((qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange)
                if rng.integers(0, 3) == 0:      # sometimes a tight vertical bound, as frame-parallel row lag makes
                    mvmax = (mvmax[0], min(mvmax[1], int(rng.integers(0, 6))))
                mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(int(rng.integers(0, 5)))]
                qp = int(rng.choice([22, 28, 37]))
                a = o.motion_estimate(refp, srcp, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp)
                b = r.motion_estimate(refp, srcp, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp)
                assert a == b, (depth, method, subme, w, h, bx, by, qmvp, mvmin, mvmax, mvc, a, b)
                n += 1
    assert n >= 60


@pytest.mark.parametrize("depth", DEPTHS)
def test_sea_search_matches_reference(depth):
    """--me sea: the restated window-sum planes against the reference's own integral_inith / integral_initv primitives driven as
    FrameFilter::computeMEIntegral drives them, and the restated search against the real MotionEstimate::motionEstimate with X265_SEA on those
    planes (shapes whose sub-blocks lie inside the PU: the reference reads its 64x64 source cache beyond the PU for the others)."""
    _need_ref(depth)
    o, r = Orc(depth), Ref(depth)
    rng = np.random.default_rng(177 + depth)
    refp, srcp, m = me_scene(depth, 99 + depth)
    H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
    pady, padx = 80, 96                                     # PicYuv margins at CTU 64 (picyuv.cpp:87-88)
    refp = np.ascontiguousarray(np.pad(refp[m:m + H, m:m + W], ((pady, pady), (padx, padx)), mode="edge"))
    srcp = np.ascontiguousarray(np.pad(srcp[m:m + H, m:m + W], ((pady, pady), (padx, padx)), mode="edge"))
    po_, pr = o.integral_planes(refp), r.integral_planes(refp, (pady, padx))
    for k, (w, h) in enumerate(Orc.SEA_WINDOWS):
        assert np.array_equal(po_[k][:refp.shape[0] - h - 1, :refp.shape[1] - w], pr[k][:refp.shape[0] - h - 1, :refp.shape[1] - w]), (k, w, h)
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (32, 24), (24, 32), (64, 48), (48, 64), (64, 16), (16, 64),
             (16, 12), (12, 16), (16, 4), (4, 16)]
    n = 0
    for subme in (0, 2, 3):
        for (w, h) in sizes:
            for _ in range(2):
                bx = padx + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = pady + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                merange = int(rng.choice([8, 16, 24]))
                qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                mvmin = ((qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange)
                mvmax = ((qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange)
                mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(int(rng.integers(0, 4)))]
                qp = int(rng.choice([22, 28, 37]))
                a = o.motion_estimate_sea(refp, srcp, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, subme, qp, planes=po_)
                b = r.motion_estimate_sea(refp, srcp, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, subme, qp, planes=pr, pad=(pady, padx))
                assert a == b, (depth, subme, w, h, bx, by, qmvp, mvmin, mvmax, mvc, a, b)
                n += 1
    assert n >= 100


@pytest.mark.parametrize("depth", DEPTHS)
def test_lowres_pass_matches_reference(depth):
    """Lowres::create/init + LookaheadTLD::lowresIntraEstimate of the real reference vs the restatement: the four hpel planes
    with their borders, every block's intra cost and mode, the row sums and the frame estimate."""
    _need_ref(depth)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.lowres_results(Orc, depth), make_golden.lowres_results(Ref, depth)
    assert set(a) == set(b)
    for k in a:
        assert same(a[k], b[k]), k


@pytest.mark.parametrize("depth", DEPTHS)
def test_intra_filter_flags_match_reference(depth):
    _need_ref(depth)
    o, r = Orc(depth), Ref(depth)
    for n in (4, 8, 16, 32):
        for mode in range(35):
            assert o.intra_uses_filtered(n, mode) == r.intra_uses_filtered(n, mode), (n, mode)


@pytest.mark.parametrize("depth", DEPTHS)
def test_lookahead_p_cost_matches_reference(depth):
    """The real Lookahead / CostEstimateGroup::estimateCUCost (lowres motionEstimate, reverse-order MV prediction, intra
    fallback, frame score) over whole frames, serial and in cooperative row slices, vs the restatement."""
    _need_ref(depth)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.lookahead_results(Orc, depth), make_golden.lookahead_results(Ref, depth)
    assert set(a) == set(b)
    for k in a:
        assert same(a[k], b[k]), k


@pytest.mark.parametrize("depth", DEPTHS)
def test_chroma_motion_estimate_matches_reference(depth):
    """The real MotionEstimate with bChromaSATD (subpelCompare's Cb + Cr SATD term, motion.cpp:1601-1660) vs the restatement:
    every PU shape x DIA / HEX / STAR x subme 2..7."""
    _need_ref(depth)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.chroma_me_results(Orc, depth), make_golden.chroma_me_results(Ref, depth)
    assert a == b, [k for k in a if a[k] != b[k]][:10]
    assert len(a) == 360


@pytest.mark.parametrize("depth", DEPTHS)
def test_lookahead_b_cost_matches_reference(depth):
    """estimateCUCost for a B frame through the real Lookahead (both lists searched or list 0 reused, bidir average and co-located
    candidates, the skip rule, 100/130 frame scaling), serial and sliced, vs the restatement."""
    _need_ref(depth)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.lookahead_b_results(Orc, depth), make_golden.lookahead_b_results(Ref, depth)
    for k in a:
        assert same(a[k], b[k]), k


@pytest.mark.parametrize("depth", DEPTHS)
def test_bipred_matches_reference(depth):
    """Predict::predInterLumaShort / predInterChromaShort of two references + Yuv::addAvg (the unweighted bi-pred branch of
    Predict::motionCompensation) vs the restatement, every PU shape."""
    _need_ref(depth)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    a, b = make_golden.bipred_results(Orc, depth), make_golden.bipred_results(Ref, depth)
    for k in a:
        assert same(a[k], b[k]), k
