"""Generate tests/golden/primitives_golden.json: sha256 digests of the REAL reference's outputs
(oracle/_ref/libx265ref{8,10}.so, i.e. /root/reference/source compiled by oracle/Makefile) on the seeded cases of
tests/cases.py.  Run here (where /root/reference exists): `python tests/golden/make_golden.py`.
tests/test_oracle_golden.py then pins the oracle to these digests anywhere, with or without oracle/_ref."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
from backends import Ref  # noqa: E402
from cases import gen_cases, coef_cases, loop_cases, deblock_cases, weight_scenes, aq_cases, cutree_cases, umh_groups, me_scene, me_scene_yuv, lowres_scene, lookahead_scene, lookahead_scene3, digest  # noqa: E402

ME_CASES = [  # (method, subme, w, h, bx_off, by_off, merange, qmvp, mvc, qp)
    (1, 2, 16, 16, 16, 24, 57, (5, -7), [(12, 8), (-20, 4)], 28),
    (1, 2, 64, 64, 32, 32, 57, (0, 0), [], 28),
    (1, 5, 8, 8, 40, 12, 32, (-13, 22), [(3, 3)], 22),
    (1, 1, 32, 24, 8, 48, 57, (9, 1), [], 37),
    (0, 2, 16, 8, 60, 60, 16, (2, 2), [(0, 8)], 28),
    (5, 2, 8, 16, 20, 44, 12, (-4, 6), [], 28),
    (5, 3, 32, 32, 64, 16, 10, (0, 0), [(16, -16)], 22),
    (1, 7, 48, 64, 0, 0, 57, (31, -29), [(8, 8), (-8, -8), (40, 0)], 28),
    (3, 3, 16, 16, 16, 24, 57, (37, -41), [(12, 8)], 28),        # X265_STAR_SEARCH (BASELINE.json configs[2])
    (3, 3, 64, 64, 32, 32, 57, (0, 0), [], 28),
    (3, 2, 8, 8, 40, 12, 32, (-53, 22), [(3, 3)], 22),
    (3, 4, 32, 32, 64, 16, 16, (60, 60), [], 37),
]


def me_digests(backend_cls, depth):
    b = backend_cls(depth)
    refp, srcp, m = me_scene(depth, 99 + depth)
    out = {}
    for i, (method, subme, w, h, bx, by, mr, qmvp, mvc, qp) in enumerate(ME_CASES):
        mvmin = ((qmvp[0] >> 2) - mr, (qmvp[1] >> 2) - mr)
        mvmax = ((qmvp[0] >> 2) + mr, (qmvp[1] >> 2) + mr)
        cost, mv = b.motion_estimate(refp, srcp, m + bx, m + by, w, h, mvmin, mvmax, qmvp, mvc, mr, method, subme, qp)
        out["me#%d" % i] = [int(cost), int(mv[0]), int(mv[1])]
    return out


def umh_results(backend_cls, depth):
    """X265_UMH_SEARCH (method 2) over tests/cases.py umh_groups: one [cost, mvx, mvy] per PU, keyed scene/group/PU."""
    b = backend_cls(depth)
    out = {}
    for gi, (si, ref, src, g) in enumerate(umh_groups(depth)):
        for i in range(len(g["pus"])):
            cost, mv = b.motion_estimate(ref, src, g["pus"][i][0], g["pus"][i][1], g["w"], g["h"], g["mins"][i], g["maxs"][i], g["mvps"][i],
                                         g["cands"][i], g["merange"], 2, g["subme"], g["qp"])
            out["umh#%d.%d.%d" % (si, gi, i)] = [int(cost), int(mv[0]), int(mv[1])]
    return out


def chroma_me_cases():
    """(w, h, bx, by, method, subme, qmvp, mvc): every PU shape x {DIA, HEX, STAR} x subme {2,3,4,5,7}, seeded positions."""
    from backends import PU_SIZES
    rng = np.random.default_rng(9)
    out = []
    for (w, h) in PU_SIZES:
        if (w, h) == (4, 4):
            continue
        for method in (0, 1, 3):
            for subme in (2, 3, 4, 5, 7):
                bx, by = int(rng.integers(0, 192 - w) // 2 * 2), int(rng.integers(0, 160 - h) // 2 * 2)
                qmvp = (int(rng.integers(-20, 21)), int(rng.integers(-20, 21)))
                mvc = [(int(rng.integers(-30, 31)), int(rng.integers(-30, 31)))] if rng.integers(0, 2) else []
                out.append((w, h, bx, by, method, subme, qmvp, mvc))
    return out


def chroma_me_results(backend_cls, depth, stride=1):
    """motionEstimate with the chroma SATD term of subpelCompare (subme > 2, 4:2:0): [cost, mvx, mvy] per case."""
    b = backend_cls(depth)
    ref, src, m = me_scene_yuv(depth, 321 + depth)
    out = {}
    for i, (w, h, bx, by, method, subme, qmvp, mvc) in enumerate(chroma_me_cases()[::stride]):
        mr = 16
        mvmin = ((qmvp[0] >> 2) - mr, (qmvp[1] >> 2) - mr)
        mvmax = ((qmvp[0] >> 2) + mr, (qmvp[1] >> 2) + mr)
        cost, mv = b.motion_estimate_chroma(ref, src, m + bx, m + by, w, h, mvmin, mvmax, qmvp, mvc, mr, method, subme, 28)
        out["cme#%d" % i] = [int(cost), int(mv[0]), int(mv[1])]
    return out


def bipred_results(backend_cls, depth):
    """Bi-predictive motion compensation (predInterLumaShort / predInterChromaShort of both references + addAvg) for every PU shape,
    six vector pairs each (full-pel, half and quarter fractions in either direction): (Y, Cb, Cr) per case."""
    from backends import PU_SIZES
    b = backend_cls(depth)
    ref, src, m = me_scene_yuv(depth, 55 + depth)
    rng = np.random.default_rng(3)
    out = {}
    for (w, h) in PU_SIZES:
        if (w, h) == (4, 4):
            continue
        for t in range(6):
            bx, by = m + int(rng.integers(0, 192 - w) // 2 * 2), m + int(rng.integers(0, 160 - h) // 2 * 2)
            mv0 = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            mv1 = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            if t == 0:
                mv0 = (mv0[0] & ~3, mv0[1] & ~3)
            if t == 1:
                mv1 = (mv1[0] & ~7, mv1[1])
            if t == 2:
                mv0 = (mv0[0], mv0[1] & ~7)
            out["bi %dx%d #%d" % (w, h, t)] = b.pred_inter_bi(ref, src, bx, by, w, h, mv0, mv1)
    return out


def coef_digests(backend_cls):
    """The coefficient-scan cost primitives (scanPosLast … costC1C2Flag) over tests/cases.py coef_cases, plus the scan orders themselves."""
    b = backend_cls(8)
    out = {label: digest(getattr(b, fn)(*args)) for label, fn, args in coef_cases()}
    for t in range(3):
        for log2 in (2, 3, 4, 5):
            out["scanOrder t%d log2 %d" % (t, log2)] = digest(b.scan_order(t, log2))
    return out


def loop_digests(backend_cls, depth):
    """The in-loop filter primitives (deblocking edge filters, SAO offset application and statistics) over tests/cases.py loop_cases."""
    b = backend_cls(depth)
    out = {label: digest(getattr(b, fn)(*args)) for label, fn, args in loop_cases(depth)}
    out.update({label: digest(getattr(b, fn)(*args)) for label, fn, args in deblock_cases(depth)})      # the real Deblock::edgeFilterLuma / Chroma
    return out


def weightp_results(backend_cls, depth):
    """LookaheadTLD::weightsAnalyse per scene of tests/cases.py weight_scenes: (isWeighted, the four weighted lowres planes)."""
    b = backend_cls(depth)
    out = {}
    for label, s0, s1, m, H, W, st in weight_scenes(depth):
        isw, planes = b.weights_analyse(s0, s1, (m, m), W, H, m, m, *st)
        out["weightp " + label] = (isw, planes[0], planes[1], planes[2], planes[3])
    return out


def lookahead_weightp_results(backend_cls, depth):
    """The P-frame cost pass with --weightp (weightsAnalyse, then list 0 searched on the weighted planes) per weight scene:
    lookahead_cost_p's tuple + isWeighted."""
    b = backend_cls(depth)
    return {"lookahead weightp " + label: b.lookahead_cost_p_weightp(s0, s1, (m, m), W, H, m, m, st) for label, s0, s1, m, H, W, st in weight_scenes(depth)}


def aq_results(backend_cls, depth):
    """LookaheadTLD::calcAdaptiveQuantFrame per case of tests/cases.py aq_cases: (blockCount, qpAqOffset, invQscaleFactor, invQscaleFactor8x8, wpStats)."""
    b = backend_cls(depth)
    return {c[0]: b.aq_frame(*c[1:]) for c in aq_cases(depth)}


def cutree_results(backend_cls):
    """Lookahead::estimateCUPropagate (+ cuTreeFinish where the backend has it) per case of tests/cases.py cutree_cases."""
    b = backend_cls(8)
    return {label: b.cutree_propagate(*args) for label, args in cutree_cases()}


def mc_cases(depth):
    """Predict::motionCompensation cases: (label, w, h, bx, by, mv0, mv1 or None, wp0, wp1, sliceP, uniList); every branch of
    predict.cpp:77-266 — P / B-uni from either list / bi, each with weighted prediction off, on-but-absent for the reference, and
    present (weights and offsets over the ranges weightAnalyse produces, denominators 0..7, all three planes different)."""
    from backends import PU_SIZES
    rng = np.random.default_rng(31 + depth)
    out = []

    def wp(present):
        denom = int(rng.integers(0, 8))
        return [(int(rng.integers(max(1, (1 << denom) // 2), min(127, 2 * (1 << denom)) + 1)), int(rng.integers(-40, 41)), denom, present)] + \
               [(int(rng.integers(max(1, (1 << d) // 2), min(127, 2 * (1 << d)) + 1)), int(rng.integers(-20, 21)), d, int(rng.integers(0, 2)))
                for d in (int(rng.integers(0, 8)), int(rng.integers(0, 8)))]
    for (w, h) in PU_SIZES:
        if (w, h) == (4, 4):
            continue
        for t in range(10):
            bx, by = int(rng.integers(0, 192 - w) // 2 * 2), int(rng.integers(0, 160 - h) // 2 * 2)
            mv0 = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            mv1 = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            if t % 5 == 0:
                mv0 = (mv0[0] & ~7, mv0[1] & ~7)
            if t % 5 == 1:
                mv0, mv1 = (mv0[0] & ~7, mv0[1]), (mv1[0], mv1[1] & ~7)
            kind = t % 10
            if kind == 0:
                out.append(("P plain", w, h, bx, by, mv0, None, None, None, 1, 0))
            elif kind == 1:
                out.append(("P weighted", w, h, bx, by, mv0, None, wp(1), None, 1, 0))
            elif kind == 2:
                out.append(("P absent", w, h, bx, by, mv0, None, wp(0), None, 1, 0))
            elif kind == 3:
                out.append(("B uni0 weighted", w, h, bx, by, mv0, None, wp(1), None, 0, 0))
            elif kind == 4:
                out.append(("B uni1 weighted", w, h, bx, by, mv0, None, wp(1), None, 0, 1))
            elif kind == 5:
                out.append(("B uni1 plain", w, h, bx, by, mv0, None, None, None, 0, 1))
            elif kind == 6:
                out.append(("B bi plain", w, h, bx, by, mv0, mv1, None, None, 0, 0))
            elif kind == 7:
                out.append(("B bi weighted", w, h, bx, by, mv0, mv1, wp(1), wp(1), 0, 0))
            elif kind == 8:
                out.append(("B bi one-sided", w, h, bx, by, mv0, mv1, wp(0), wp(1), 0, 0))
            else:
                out.append(("B bi absent", w, h, bx, by, mv0, mv1, wp(0), wp(0), 0, 0))
    return out


def mc_results(backend_cls, depth):
    b = backend_cls(depth)
    ref, src, m = me_scene_yuv(depth, 58 + depth)
    out = {}
    for i, (label, w, h, bx, by, mv0, mv1, wp0, wp1, sliceP, uniList) in enumerate(mc_cases(depth)):
        out["mc %s %dx%d #%d" % (label, w, h, i)] = b.motion_compensation(ref, src if mv1 is not None else None, m + bx, m + by, w, h, mv0, mv1,
                                                                           wp0, wp1, sliceP, uniList)
    return out


LOWRES_CASES = [(200, 136), (176, 144), (66, 50)]   # (W, H) of the full-resolution picture


def lowres_results(backend_cls, depth):
    """Lowres::init + LookaheadTLD::lowresIntraEstimate per case: (costEst, intraCost, intraMode, rowSatds, planes)."""
    b = backend_cls(depth)
    out = {}
    for i, (w, h) in enumerate(LOWRES_CASES):
        src, m = lowres_scene(depth, 300 + depth + i, h, w)
        est, cost, mode, rows, planes, (stride, lw, lh) = b.lowres_pass(src, (m, m), w, h, m, m)
        if i == 0:   # the rows to the right of the picture margin are allocator padding in the reference: compare the defined area
            assert stride >= lw + 2 * m
        out["lowres#%d" % i] = (est, cost, mode, rows) + tuple(np.ascontiguousarray(p[:, :lw + 2 * m]) for p in planes)
    return out


def lowres_digests(backend_cls, depth):
    return {k: digest(v) for k, v in lowres_results(backend_cls, depth).items()}


LOOKAHEAD_CASES = [(200, 136, 9, 1), (200, 136, 3, 3), (176, 144, 9, 1), (66, 50, 4, 1), (320, 200, 5, 2)]   # W, H, rows/slice, slices


def lookahead_results(backend_cls, depth):
    """The lookahead's P-frame cost pass per case: (costEst, mvs, mvCosts, lowresCosts, rowSatds, intraMbs, intraCost)."""
    b = backend_cls(depth)
    out = {}
    for i, (w, h, rps, ns) in enumerate(LOOKAHEAD_CASES):
        s0, s1, m = lookahead_scene(depth, 700 + depth + w, h, w)
        out["lookahead#%d" % i] = b.lookahead_cost_p(s0, s1, (m, m), w, h, m, m, rps, ns)
    return out


LOOKAHEAD_B_CASES = [(200, 136, 9, 1, 0), (200, 136, 9, 1, 1), (200, 136, 3, 3, 0), (176, 144, 9, 1, 0), (320, 200, 5, 2, 1), (66, 50, 4, 1, 0)]


def lookahead_b_results(backend_cls, depth):
    """B-frame cost pass per case (W, H, rows/slice, slices, list 0 pre-searched): (costEst, mvs0, mvCosts0, mvs1, mvCosts1, lowresCosts, rowSatds)."""
    b = backend_cls(depth)
    out = {}
    for i, (w, h, rps, ns, pre) in enumerate(LOOKAHEAD_B_CASES):
        pics, m = lookahead_scene3(depth, 800 + depth + w, h, w)
        out["lookahead_b#%d" % i] = b.lookahead_cost_b(pics[0], pics[1], pics[2], (m, m), w, h, m, m, rps, ns, pre)
    return out


def lookahead_digests(backend_cls, depth):
    return {k: digest(v) for k, v in lookahead_results(backend_cls, depth).items()}


def prim_digests(backend_cls, depth):
    b = backend_cls(depth)
    out = {}
    for label, fn, args in gen_cases(depth):
        key = label
        k = 0
        while key in out:
            k += 1
            key = "%s~%d" % (label, k)
        out[key] = digest(getattr(b, fn)(*args))
    return out


SEA_SHAPES = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (32, 24), (24, 32), (64, 48), (48, 64), (64, 16), (16, 64),
              (16, 12), (12, 16), (16, 4), (4, 16)]


def sea_results(B, depth):
    """--me sea on fixed cases (the shapes whose sub-blocks lie inside the PU): [cost, mvx, mvy] per case + a digest of the twelve window-sum planes
    over the region where the reference defines them."""
    from cases import me_scene
    b = B(depth)
    rng = np.random.default_rng(977 + depth)
    refp, srcp, m = me_scene(depth, 299 + depth)
    H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
    pady, padx = 80, 96
    refp = np.ascontiguousarray(np.pad(refp[m:m + H, m:m + W], ((pady, pady), (padx, padx)), mode="edge"))
    srcp = np.ascontiguousarray(np.pad(srcp[m:m + H, m:m + W], ((pady, pady), (padx, padx)), mode="edge"))
    is_ref = B is Ref
    planes = b.integral_planes(refp, (pady, padx)) if is_ref else b.integral_planes(refp)
    wins = ((32, 32), (32, 24), (32, 8), (24, 32), (16, 16), (16, 12), (16, 4), (12, 16), (8, 32), (8, 8), (4, 16), (4, 4))
    out = {"planes": digest(tuple(np.ascontiguousarray(planes[k][:refp.shape[0] - h - 1, :refp.shape[1] - w]) for k, (w, h) in enumerate(wins))), "cases": []}
    for subme in (0, 2, 3):
        for (w, h) in SEA_SHAPES:
            bx = padx + int(rng.integers(0, (W - w) // 4 + 1)) * 4
            by = pady + int(rng.integers(0, (H - h) // 4 + 1)) * 4
            merange = int(rng.choice([8, 16, 24]))
            qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
            mvmin = ((qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange)
            mvmax = ((qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange)
            mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(int(rng.integers(0, 4)))]
            qp = int(rng.choice([22, 28, 37]))
            kw = dict(planes=planes, pad=(pady, padx)) if is_ref else dict(planes=planes)
            c, mv = b.motion_estimate_sea(refp, srcp, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, subme, qp, **kw)
            out["cases"].append([int(c), int(mv[0]), int(mv[1])])
    return out


if __name__ == "__main__":
    gold = {}
    for depth in (8, 10, 12):
        gold[str(depth)] = {"prims": prim_digests(Ref, depth), "me": me_digests(Ref, depth), "umh": umh_results(Ref, depth), "chroma_me": chroma_me_results(Ref, depth),
                            "bipred": {k: digest(v) for k, v in bipred_results(Ref, depth).items()},
                            "mc": {k: digest(v) for k, v in mc_results(Ref, depth).items()},
                            "loop": loop_digests(Ref, depth),
                            "weightp": {k: digest(v) for k, v in weightp_results(Ref, depth).items()},
                            "aq": {k: digest(v) for k, v in aq_results(Ref, depth).items()},
                            "lookahead_weightp": {k: digest(v) for k, v in lookahead_weightp_results(Ref, depth).items()}, "lowres": lowres_digests(Ref, depth), "lookahead": lookahead_digests(Ref, depth),
                            "lookahead_b": {k: digest(v) for k, v in lookahead_b_results(Ref, depth).items()},
                            "mvcost": {str(qp): digest(Ref(depth).mvcost_table(qp)) for qp in (12, 28, 37, 51)},
                            "sea": sea_results(Ref, depth)}
    # the CABAC cost table is data of the reference: dump it for the tests and for the GPU box, where /root/reference does not exist
    with open(os.path.join(HERE, "entropy_state_bits.json"), "w") as f:
        json.dump({"source": "x265_entropyStateBits (common/constants.cpp) of the reference build, dumped by tests/golden/make_golden.py",
                   "entropyStateBits": [int(v) for v in Ref(8).entropy_state_bits()]}, f)
    gold["coef"] = coef_digests(Ref)
    gold["cutree"] = {k: [digest(v[0]), digest(v[1]), digest(v[2])] for k, v in cutree_results(Ref).items()}
    path = os.path.join(HERE, "primitives_golden.json")
    with open(path, "w") as f:
        json.dump({"source": "x265 3.4+28 C primitives ([noasm]), /root/reference/source via oracle/Makefile",
                   "generator": "tests/golden/make_golden.py", "golden": gold}, f, indent=0, sort_keys=True)
    print("wrote", path, {d: len(g["prims"]) for d, g in gold.items() if "prims" in g})
