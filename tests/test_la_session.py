"""The lookahead SESSION of libx265hip.so (x265hip_la_*, x265_amd/csrc/lasession.hip) against the same ABI implemented on the oracle
(tests/support/libx265hip_emul.so, test infrastructure): both libraries are driven through an identical randomised script — frames entering and
leaving slots, weighted-prediction analyses on fades, estimate batches mixing P and B estimates with and without searches, cooperative slices,
vector uploads (x265hip_la_put_vectors), searches handed over ahead of their request (x265hip_la_search) in every flavour, then requests that are
served from them — and every output array of every call must be identical.  This is the device path the timed encode spends its device time
in; a failure here localises what would otherwise show up as "bitstreams differ" (tests/test_x265_dropin.py).  Reference behaviour:
CostEstimateGroup::estimateFrameCost / estimateCUCost, source/encoder/slicetype.cpp:3115-3385 (pinned on the oracle side by
tests/test_oracle_vs_ref.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

EMUL = os.path.join(ROOT, "tests", "support", "libx265hip_emul.so")


def _libs():
    import x265_amd.hipprim as hp
    hip = hp.lib()
    if not os.path.exists(EMUL):
        pytest.skip("tests/support/libx265hip_emul.so not built (make -C oracle emul)")
    em = C.CDLL(EMUL)
    for name, (res, args) in hp.PROTOTYPES.items():
        if name.startswith("x265hip_la_") or name in ("x265hip_last_error",):
            fn = getattr(em, name)
            fn.restype, fn.argtypes = res, args
    return hp, hip, em


def _frames(depth, seed, count, W, H, margin, fade):
    """`count` padded pictures: per-tile constant motion, noise; with `fade` the brightness ramps (weightp finds weights)."""
    from cases import textured_frame
    rng = np.random.default_rng(seed)
    pmax = (1 << depth) - 1
    big = textured_frame(rng, H + 128, W + 128, depth, sigma=2.0)
    th, tw = 40, 48
    vec = {(y0, x0): (int(rng.integers(-5, 6)), int(rng.integers(-5, 6))) for y0 in range(0, H, th) for x0 in range(0, W, tw)}
    pics, stats = [], []
    for t in range(count):
        p = np.zeros((H, W), np.float64)
        for (y0, x0), (dy, dx) in vec.items():
            y1, x1 = min(y0 + th, H), min(x0 + tw, W)
            p[y0:y1, x0:x1] = big[64 + y0 + t * dy:64 + y1 + t * dy, 64 + x0 + t * dx:64 + x1 + t * dx]
        if fade:
            p = p * (0.45 + 0.55 * t / max(count - 1, 1)) + 3 * t
        p = np.clip(np.rint(p + rng.normal(0, 2.0 * (pmax / 255.0), p.shape)), 0, pmax).astype(big.dtype)
        core = p.astype(np.int64)
        n = core.size
        sm, sq = int(core.sum()), int((core * core).sum())
        stats.append(((sq - (sm * sm + n // 2) // n) // 4, sm // 4))          # wp_ssd[0], wp_sum[0] at the scale weightsAnalyse expects (cases.weight_scenes)
        pics.append(np.ascontiguousarray(np.pad(p, ((margin, margin), (margin, margin + 8)), mode="edge")))
    return pics, stats


class _Side:
    """One library driving one session; outputs are kept per call for comparison."""

    def __init__(self, hp, L, cfg, ncu, hcu):
        self.hp, self.L, self.ncu, self.hcu = hp, L, ncu, hcu
        self.la = L.x265hip_la_create(C.byref(cfg))
        assert self.la, L.x265hip_last_error()

    def close(self):
        self.L.x265hip_la_destroy(self.la)

    def ok(self, code):
        assert code == 0, self.L.x265hip_last_error()


CONFIGS = [
    # depth, W, H, frames, bframes, fade, seed
    (8, 200, 136, 7, 3, False, 11),
    (8, 320, 200, 6, 2, True, 12),
    (10, 176, 144, 6, 3, True, 13),
    (12, 200, 136, 5, 2, False, 14),
    (8, 66, 50, 5, 2, False, 15),
    (8, 400, 300, 8, 4, False, 16),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfgi", range(len(CONFIGS)))
def test_session_matches_emulation(cfgi):
    hp, hip, em = _libs()
    _drive(hp, hip, em, cfgi)


def test_session_script_is_deterministic_on_the_emulation():
    """CPU tier: the same script with the emulation on both sides — keeps the script itself (slot moves, flags, flavours) honest without a GPU."""
    hp, _, em = _libs()
    _drive(hp, em, em, 0)


def _drive(hp, hip, em, cfgi):
    from backends import Orc
    depth, W, H, count, bframes, fade, seed = CONFIGS[cfgi]
    margin = 80
    pics, stats = _frames(depth, seed, count, W, H, margin, fade)
    o = Orc(depth)
    low = []
    for p in pics:
        _, icost, _, _, planes, (stride, lw, lh) = o.lowres_pass(p, (margin, margin), W, H, margin, margin)
        low.append((np.ascontiguousarray(np.concatenate([pl.reshape(-1) for pl in planes])), np.ascontiguousarray(icost)))
    wcu, hcu = lw // 8, lh // 8
    ncu = wcu * hcu
    planeElems = (lh + 2 * margin) * stride
    cfg = hp.LaConfig(depth, lw, lh, stride, planeElems, margin * stride + margin, wcu, hcu, bframes + 2, count + 2)
    rng = np.random.default_rng(1000 + seed)
    invq = [np.ascontiguousarray(rng.integers(180, 330, ncu).astype(np.int32)) if k % 3 != 2 else None for k in range(count)]
    sides = [_Side(hp, hip, cfg, ncu, hcu), _Side(hp, em, cfg, ncu, hcu)]
    # slot permutation + two spare slots: a frame may move to another slot (eviction) in the middle of the script
    slot_of = {k: int(s) for k, s in enumerate(rng.permutation(count + 2)[:count])}
    free = [s for s in range(count + 2) if s not in slot_of.values()]

    def set_frame(k):
        for sd in sides:
            sd.ok(sd.L.x265hip_la_set_frame(sd.la, slot_of[k], low[k][0].ctypes.data, low[k][1].ctypes.data, invq[k].ctypes.data if invq[k] is not None else None))

    for k in range(count):
        set_frame(k)
    host_mvs = {}            # (frame, list, dist) -> (mvs, mvCosts) as the host's Lowres would hold them after the first search

    def geom(coop):
        if not coop or hcu < 4:
            return hcu, 1
        slices = 2 + int(rng.integers(0, 3))
        rows = max(hcu // slices, 1)
        while rows * (slices - 1) >= hcu:
            slices -= 1
        return rows, slices

    def run_batch(jobs, rows, slices, ahead):
        """jobs: list of (p0, p1, b).  Returns nothing; asserts both sides agree on everything they return."""
        results = []
        for sd in sides:
            est = (hp.LaEstimate * max(len(jobs), 1))()
            keep = []
            pending = set()          # searched by an earlier estimate of this batch: later ones reuse it (the reference never queues a search twice)
            for i, (p0, p1, b) in enumerate(jobs):
                e = est[i]
                e.b, e.p0, e.p1 = slot_of[b], slot_of[p0], slot_of[p1]
                e.dist0, e.dist1 = b - p0, p1 - b
                e.search0 = int((b, 0, b - p0) not in host_mvs and (b, 0, b - p0) not in pending)
                e.search1 = int(p1 > b and (b, 1, p1 - b) not in host_mvs and (b, 1, p1 - b) not in pending)
                pending.add((b, 0, b - p0))
                if p1 > b:
                    pending.add((b, 1, p1 - b))
                e.weightedId = -1
                if fade and e.search0 and not sd.L.x265hip_la_has_ahead(sd.la, e.b, 0, e.dist0, int(p1 > b), rows, slices):
                    chosen, isw, wid = hp.WeightParam(), C.c_int(0), C.c_int(-1)
                    sd.ok(sd.L.x265hip_la_weights_analyse(sd.la, e.b, e.p0, stats[b][0], stats[b][1], stats[p0][0], stats[p0][1], C.byref(chosen), C.byref(isw), C.byref(wid)))
                    e.weightedId = wid.value
                    keep.append(("w", isw.value, chosen.inputWeight, chosen.inputOffset, chosen.log2WeightDenom))
                arrs = [np.full(2 * ncu, -7, np.int32), np.full(ncu, -7, np.int32), np.full(2 * ncu, -7, np.int32), np.full(ncu, -7, np.int32),
                        np.full(ncu, 0xABCD, np.uint16), np.full(hcu, -7, np.int32)]
                e.mvs0, e.mvCosts0, e.mvs1, e.mvCosts1, e.lowresCosts, e.rowSatds = [a.ctypes.data for a in arrs]
                keep.append(arrs)
            ah = (hp.LaSearch * max(len(ahead), 1))()
            for j, (b, ref, lst, bidir, arows, aslices) in enumerate(ahead):
                a = ah[j]
                a.b, a.ref, a.list, a.dist, a.bidir, a.weightedId = slot_of[b], slot_of[ref], lst, abs(b - ref), bidir, -1
                a.numRowsPerSlice, a.numSlices = arows, aslices
                if fade and lst == 0:
                    chosen, isw, wid = hp.WeightParam(), C.c_int(0), C.c_int(-1)
                    sd.ok(sd.L.x265hip_la_weights_analyse(sd.la, a.b, a.ref, stats[b][0], stats[b][1], stats[ref][0], stats[ref][1], C.byref(chosen), C.byref(isw), C.byref(wid)))
                    a.weightedId = wid.value
            sd.ok(sd.L.x265hip_la_estimate_batch_ahead(sd.la, est, len(jobs), rows, slices, ah, len(ahead)))
            results.append(([(e.costEst, e.costEstAq, e.intraMbs, e.search0, e.search1) for e in est[:len(jobs)]], keep))
        (sc0, k0), (sc1, k1) = results
        assert sc0 == sc1, (jobs, sc0, sc1)
        for a, b_ in zip(k0, k1):
            if isinstance(a, tuple):
                assert a == b_, (jobs, a, b_)
                continue
            for x, y in zip(a, b_):
                assert np.array_equal(x, y), jobs
        # what the host would now hold
        it = iter([k for k in k0 if not isinstance(k, tuple)])
        for (p0, p1, b), (_, _, _, s0, s1) in zip(jobs, sc0):
            arrs = next(it)
            if s0:
                host_mvs[(b, 0, b - p0)] = (arrs[0].copy(), arrs[1].copy())
            if s1:
                host_mvs[(b, 1, p1 - b)] = (arrs[2].copy(), arrs[3].copy())

    maxd = bframes + 1
    # 1. a batch in batch geometry: P and B estimates with both lists searched, + searches ahead in the cooperative geometry (both flavours)
    crows, cslices = geom(True)
    jobs = []
    for b in range(2, count - 1):
        d = 1 + int(rng.integers(0, min(maxd - 1, b, count - 1 - b)))
        jobs.append((b - d, b + d, b) if rng.integers(0, 2) else (b - d, b, b))
    ahead = [(1, 0, 0, 0, crows, cslices), (1, 0, 0, 1, crows, cslices)]
    for b in range(1, count - 1):
        for d in range(1, min(bframes, count - 1 - b) + 1):
            if not any(j[2] == b and j[1] == b + d for j in jobs):
                ahead.append((b, b + d, 1, 1, crows, cslices))
    last = count - 1
    for d in range(1, min(maxd, last) + 1):
        ahead.append((last, last - d, 0, 0, crows, cslices))
    run_batch(jobs, hcu, 1, ahead)
    for sd in sides:
        la, used = C.c_uint64(0), C.c_uint64(0)
        sd.ok(sd.L.x265hip_la_stats_ahead(sd.la, C.byref(la), C.byref(used), None, None))
        assert la.value == len(ahead) and used.value == 0
    # 2. single estimates in the cooperative geometry: some served from the searches ahead, some searching themselves, some reusing stored vectors
    singles = [(0, 1, 1), (0, 2, 1)] + [(last - d, last, last) for d in range(1, min(maxd, last) + 1)]
    for b in range(1, count - 1):
        for d1 in range(1, min(bframes, count - 1 - b) + 1):
            d0 = 1 + int(rng.integers(0, min(maxd - 1, b)))
            if d0 + d1 <= maxd:
                singles.append((b - d0, b + d1, b))
    for j in singles:
        run_batch([j], crows, cslices, [])
    for sd in sides:
        la, used = C.c_uint64(0), C.c_uint64(0)
        sd.ok(sd.L.x265hip_la_stats_ahead(sd.la, C.byref(la), C.byref(used), None, None))
        assert used.value >= 3, used.value
    # 3. a frame leaves its slot and comes back in another one (the binding's eviction): the session has forgotten its vectors, the host has not
    k = 2
    slot_of[k], free[0] = free[0], slot_of[k]
    set_frame(k)
    for (f, lst, d), (mv, mc) in list(host_mvs.items()):
        if f == k:
            for sd in sides:
                assert not sd.L.x265hip_la_has_vectors(sd.la, slot_of[k], lst, d)
                sd.ok(sd.L.x265hip_la_put_vectors(sd.la, slot_of[k], lst, d, mv.ctypes.data, mc.ctypes.data))
    run_batch([(k - 1, k, k), (k - 1, k + 1, k), (k - 2, k + 1, k)], hcu, 1, [])
    # 4. a batch whose later estimates reuse what an earlier estimate of the SAME batch searches (must read the producer's block)
    host_mvs.clear()
    for kk in range(count):
        set_frame(kk)
    run_batch([(1, 3, 2), (1, 2, 2), (0, 3, 2), (1, 3, 2)] if count > 3 else [(0, 1, 1), (0, 1, 1)], hcu, 1, [])
    for sd in sides:
        sd.close()
