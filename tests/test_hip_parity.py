"""GPU parity: every HIP entry point of libx265hip.so (through the C ABI) against the pinned CPU oracle, bit-exact.

The per-primitive sweep reuses the TestBench-shaped seeded cases of tests/cases.py (random / all-min / all-max
buffers, unaligned origins, stride 96) — the same cases that pin the oracle to the real reference and to the golden
digests — and reports EVERY mismatch (grouped), not just the first, so one GPU run tells the whole story."""
import collections
import os
import sys

import numpy as np
import pytest

from backends import Orc, same, PU_SIZES
from cases import gen_cases, me_scene, pix_buf, short_buf

pytestmark = pytest.mark.gpu

DEPTHS = [8, 10]
NOT_ON_GPU = set()        # every primitive of tests/cases.py has a HIP entry point


@pytest.fixture(scope="module")
def hipmod():
    import hipbackend
    from x265_amd import hipprim as hp
    assert hp.lib().x265hip_device_count() > 0, "no HIP device: the -m gpu tests need a real MI355X"
    hp.check(hp.lib().x265hip_init(0))
    return hipbackend


def _report(bad, total):
    if not bad:
        return
    groups = collections.Counter(" ".join(lbl.split()[:2]) for lbl in bad)
    msg = "%d / %d mismatches; by (primitive size): %s; first: %s" % (len(bad), total, dict(groups.most_common(40)), bad[:6])
    pytest.fail(msg)


@pytest.mark.parametrize("depth", DEPTHS)
def test_every_primitive_matches_oracle(hipmod, depth):
    o, g = Orc(depth), hipmod.Hip(depth)
    bad, n = [], 0
    for label, fn, args in gen_cases(depth):
        if fn in NOT_ON_GPU:
            continue
        want = getattr(o, fn)(*args)
        try:
            got = getattr(g, fn)(*args)
        except Exception as e:  # noqa: BLE001 - report and continue
            bad.append("%s !! %s" % (label, str(e)[:80]))
            hipmod._release()
            continue
        hipmod._release()
        n += 1
        if not same(got, want):
            bad.append(label)
    assert n > 2500
    _report(bad, n)


@pytest.mark.parametrize("depth", DEPTHS)
def test_chroma_sa8d8_and_copies(hipmod, depth):
    o, g = Orc(depth), hipmod.Hip(depth)
    rng = np.random.default_rng(5 + depth)
    a, b = pix_buf(rng, "rand", (160, 96), depth), pix_buf(rng, "rand", (160, 96), depth)
    s = short_buf(rng, "rand", (160, 96), -(1 << depth) + 1, (1 << depth) - 1)
    bad = []
    for (w, h) in [(8, 8), (16, 16), (32, 32), (8, 16), (16, 8), (16, 32), (32, 16), (8, 32), (32, 8), (24, 32), (32, 24), (16, 24)]:
        if w % 8 or h % 8:
            continue
        want = o._f("orc_sa8d8")(o_ptr(a, 3, 5), 96, o_ptr(b, 7, 2), 96, w, h)
        if g.sa8d8(w, h, a, (3, 5), b, (7, 2)) != want:
            bad.append("sa8d8 %dx%d" % (w, h))
    for (w, h) in PU_SIZES + [(2, 4), (6, 8), (2, 8), (8, 2), (8, 6), (4, 2)]:
        for kind, src in ((0, a), (1, s), (2, a), (3, s)):
            got = g.copy(kind, w, h, src, (9, 11))
            hipmod._release()
            blk = src[9:9 + h, 11:11 + w]
            want = blk.astype(got.dtype)   # sp: truncating cast like (pixel)b[x]; ps/ss: widening / identity
            if not np.array_equal(got, want):
                bad.append("copy%d %dx%d" % (kind, w, h))
    _report(bad, 1)


def o_ptr(a, y, x):
    from oracle import pyoracle as po
    return po.ptr(a, y, x)


# ---- batched launches: many jobs per call, ragged tails, all shapes ----------------------------------------------------
@pytest.mark.parametrize("depth", DEPTHS)
def test_pixcmp_batches(hipmod, depth):
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check, dev_i32
    o = Orc(depth)
    rng = np.random.default_rng(100 + depth)
    H, W = 200, 328
    a, b = pix_buf(rng, "rand", (H, W), depth), pix_buf(rng, "rand", (H, W), depth)
    da, db = DevBuf(a), DevBuf(b)
    L = hp.lib()
    bad = []
    ops = [(hp.CMP_SAD, "orc_sad", None), (hp.CMP_SATD, "orc_satd", None), (hp.CMP_SA8D, "orc_sa8d", "sq"),
           (hp.CMP_SA8D8, "orc_sa8d8", "m8"), (hp.CMP_PSY, "orc_psy_cost_pp", "sq")]
    for (w, h) in PU_SIZES:
        for op, fn, cons in ops:
            if cons == "sq" and w != h:
                continue
            if cons == "m8" and (w % 8 or h % 8):
                continue
            n = 37 if w * h >= 1024 else 301            # ragged: not a multiple of the blocks-per-wave
            ya, xa = rng.integers(0, H - h, n), rng.integers(0, W - w, n)
            yb, xb = rng.integers(0, H - h, n), rng.integers(0, W - w, n)
            oa, ob = dev_i32(ya * W + xa), dev_i32(yb * W + xb)
            out = DevBuf.zeros((n,), np.int32)
            check(L.x265hip_pixcmp_batch(op, depth, w, h, da.ptr, W, db.ptr, W, oa.ptr, ob.ptr, n, out.ptr, None))
            got = out.get()
            f = o._f(fn)
            for i in range(n):
                pa, pb = o_ptr(a, int(ya[i]), int(xa[i])), o_ptr(b, int(yb[i]), int(xb[i]))
                want = f(pa, W, pb, W, w) if fn in ("orc_sa8d", "orc_psy_cost_pp") else f(pa, W, pb, W, w, h)
                if int(got[i]) != want:
                    bad.append("%s %dx%d job%d got %d want %d" % (fn[4:], w, h, i, got[i], want))
                    break
    _report(bad, 1)


@pytest.mark.parametrize("depth", DEPTHS)
def test_transform_batches(hipmod, depth):
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check, dev_i32
    from oracle import pyoracle as po
    O = po.oracle()
    L = hp.lib()
    rng = np.random.default_rng(200 + depth)
    pmax = (1 << depth) - 1
    bad = []
    for size in (4, 8, 16, 32):
        for n in (1, 7, 67, 530):
            log2n = size.bit_length() - 1
            S = 1100                                             # residual plane stride
            plane = short_buf(rng, "rand", (size * 2 + 40, S), -pmax, pmax)
            ys, xs = rng.integers(0, 40, n), rng.integers(0, S - size, n)
            offs = dev_i32(ys * S + xs)
            dp = DevBuf(plane)
            dst = DevBuf.zeros((n, size * size), np.int16)
            check(L.x265hip_dct_batch(size, 0, depth, dp.ptr, S, offs.ptr, dst.ptr, n, None))
            got = dst.get()
            want = np.zeros_like(got)
            for i in range(n):
                O.orc_dct(log2n, po.ptr(plane, int(ys[i]), int(xs[i])), po.ptr(want, i, 0), S, depth)
            if not np.array_equal(got, want):
                bad.append("dct %d n=%d (%d TUs differ)" % (size, n, int((got != want).any(axis=1).sum())))
            # inverse of the quantised-looking coefficients, scattered back into a plane
            lim = (1 << (depth + 4)) - 1
            coef = short_buf(rng, "rand", (n, size * size), -lim, lim)
            coef[:: 3] = short_buf(rng, "rand", coef[:: 3].shape, -32768, 32767)
            dc = DevBuf(coef)
            gx = (np.arange(n) % 30) * 36
            gy = (np.arange(n) // 30) * 33
            S2 = 30 * 36 + 8
            outp = DevBuf.zeros((int(gy.max()) + 40, S2), np.int16)
            offd = dev_i32(gy * S2 + gx)
            check(L.x265hip_idct_batch(size, 0, depth, dc.ptr, outp.ptr, S2, offd.ptr, n, None))
            gotp = outp.get()
            wantp = np.zeros_like(gotp)
            for i in range(n):
                O.orc_idct(log2n, po.ptr(coef, i, 0), po.ptr(wantp, int(gy[i]), int(gx[i])), S2, depth)
            if not np.array_equal(gotp, wantp):
                bad.append("idct %d n=%d" % (size, n))
            # quant / nquant / dequant / count_nonzero on the forward output
            nc = size * size
            qp = int(rng.integers(10, 45))
            qc = np.full(nc, [26214, 23302, 20560, 18396, 16384, 14564][qp % 6], np.int32)
            qbits = 14 + qp // 6 + (15 - depth - log2n)
            add = 171 << (qbits - 9)
            dq = DevBuf(qc)
            dcoef = DevBuf(want)
            du, ql, ns = DevBuf.zeros((n, nc), np.int32), DevBuf.zeros((n, nc), np.int16), DevBuf.zeros((n,), np.uint32)
            check(L.x265hip_quant_batch(dcoef.ptr, dq.ptr, du.ptr, ql.ptr, qbits, add, nc, n, ns.ptr, None))
            gq, gdu, gns = ql.get(), du.get(), ns.get()
            wq, wdu, wns = np.zeros_like(gq), np.zeros_like(gdu), np.zeros_like(gns)
            for i in range(n):
                wns[i] = O.orc_quant(po.ptr(want, i, 0), po.ptr(qc), po.ptr(wdu, i, 0), po.ptr(wq, i, 0), qbits, add, nc)
            if not (np.array_equal(gq, wq) and np.array_equal(gdu, wdu) and np.array_equal(gns, wns)):
                bad.append("quant %d n=%d" % (size, n))
            cnt = DevBuf.zeros((n,), np.uint32)
            check(L.x265hip_count_nonzero_batch(ql.ptr, nc, n, cnt.ptr, None))
            if not np.array_equal(cnt.get(), (wq != 0).sum(axis=1).astype(np.uint32)):
                bad.append("count_nonzero %d n=%d" % (size, n))
            shift = 20 - 14 - (15 - depth - log2n)
            scale = [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6)
            if shift >= 1:
                dd = DevBuf.zeros((n, nc), np.int16)
                check(L.x265hip_dequant_normal(ql.ptr, dd.ptr, n * nc, scale, shift, None))
                wd = np.zeros((n, nc), np.int16)
                O.orc_dequant_normal(po.ptr(wq), po.ptr(wd), n * nc, scale, shift)
                if not np.array_equal(dd.get(), wd):
                    bad.append("dequant_normal %d n=%d" % (size, n))
    _report(bad, 1)


@pytest.mark.parametrize("depth", DEPTHS)
def test_residual_chain_equals_primitive_sequence(hipmod, depth):
    """x265hip_residual_chain_batch == sub_ps -> dct -> quant -> dequant_normal -> idct -> add_ps -> sse_pp (oracle)."""
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check, dev_i32
    from oracle import pyoracle as po
    O = po.oracle()
    o = Orc(depth)
    L = hp.lib()
    rng = np.random.default_rng(300 + depth)
    bad = []
    H, W = 136, 264
    fenc = pix_buf(rng, "rand", (H, W), depth)
    noise = rng.integers(-12, 13, size=(H, W))
    pred = np.clip(fenc.astype(np.int64) + noise * (1 << (depth - 8)), 0, (1 << depth) - 1).astype(fenc.dtype)
    pred[:40] = pix_buf(rng, "rand", (40, W), depth)         # a band of large residuals
    df, dp = DevBuf(fenc), DevBuf(pred)
    for size in (4, 8, 16, 32):
        log2n = size.bit_length() - 1
        nc = size * size
        for qp in (22, 37):
            tus = [(y, x) for y in range(0, H - size + 1, size) for x in range(0, W - size + 1, size)]
            tus = tus[: len(tus) - 3]                              # ragged tail
            n = len(tus)
            offs = np.array([y * W + x for (y, x) in tus], np.int32)
            do = dev_i32(offs)
            qc = np.full(nc, [26214, 23302, 20560, 18396, 16384, 14564][qp % 6], np.int32)
            qbits = 14 + qp // 6 + (15 - depth - log2n)
            add = 85 << (qbits - 9)
            shift = 20 - 14 - (15 - depth - log2n)
            scale = [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6)
            drec = DevBuf.zeros((H, W), fenc.dtype)
            lvl, ns, dist = DevBuf.zeros((n, nc), np.int16), DevBuf.zeros((n,), np.uint32), DevBuf.zeros((n,), np.uint64)
            check(L.x265hip_residual_chain_batch(size, depth, df.ptr, W, dp.ptr, W, drec.ptr, W, do.ptr, do.ptr, do.ptr,
                                                 _keep(qc).ptr, qbits, add, scale, shift,
                                                 lvl.ptr, ns.ptr, dist.ptr, n, None))
            grec, glvl, gns, gdist = drec.get(), lvl.get(), ns.get(), dist.get()
            wrec = np.zeros_like(grec)
            for i, (y, x) in enumerate(tus):
                resi = o.sub_ps(size, fenc, (y, x), pred, (y, x))
                coef = np.zeros(nc, np.int16)
                O.orc_dct(log2n, po.ptr(resi), po.ptr(coef), size, depth)
                q, du = np.zeros(nc, np.int16), np.zeros(nc, np.int32)
                nsig = O.orc_quant(po.ptr(coef), po.ptr(qc), po.ptr(du), po.ptr(q), qbits, add, nc)
                dq = np.zeros(nc, np.int16)
                O.orc_dequant_normal(po.ptr(q), po.ptr(dq), nc, scale, shift)
                r2 = np.zeros((size, size), np.int16)
                O.orc_idct(log2n, po.ptr(dq), po.ptr(r2), size, depth)
                rec = o.add_ps(size, pred, (y, x), r2, (0, 0))
                wrec[y:y + size, x:x + size] = rec
                d = o._f("orc_sse_pp")(po.ptr(fenc, y, x), W, po.ptr(rec), size, size, size)
                if not (np.array_equal(glvl[i], q) and gns[i] == nsig and int(gdist[i]) == d):
                    bad.append("chain %d qp%d tu%d (level %s numSig %d/%d dist %d/%d)" % (
                        size, qp, i, np.array_equal(glvl[i], q), gns[i], nsig, int(gdist[i]), d))
                    break
            mask = np.zeros((H, W), bool)
            for (y, x) in tus:
                mask[y:y + size, x:x + size] = True
            if not np.array_equal(grec[mask], wrec[mask]):
                bad.append("chain-recon %d qp%d" % (size, qp))
            if grec[~mask].any():
                bad.append("chain-recon-outside %d qp%d" % (size, qp))
    _report(bad, 1)


_KEEPALIVE = []


def _keep(arr):
    from x265_amd.hipprim import DevBuf
    b = DevBuf(arr)
    _KEEPALIVE.append(b)
    return b


@pytest.mark.parametrize("depth", DEPTHS)
def test_interp_batches(hipmod, depth):
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check, dev_i32
    o = Orc(depth)
    L = hp.lib()
    rng = np.random.default_rng(400 + depth)
    H, W = 180, 300
    p = pix_buf(rng, "rand", (H, W), depth)
    sh = short_buf(rng, "rand", (H, W), -8192, 8191)
    dP, dS = DevBuf(p), DevBuf(sh)
    bad = []
    kinds = [("hpp", hp.IF_HPP), ("hps", hp.IF_HPS), ("vpp", hp.IF_VPP), ("vps", hp.IF_VPS), ("vsp", hp.IF_VSP), ("vss", hp.IF_VSS), ("hvpp", hp.IF_HVPP)]
    for (w, h) in [(8, 8), (16, 16), (64, 64), (12, 16), (32, 24), (4, 8), (48, 64)]:
        for name, k in kinds:
            n = 23
            src = sh if name in ("vsp", "vss") else p
            dsrc = dS if name in ("vsp", "vss") else dP
            ys, xs = rng.integers(8, H - h - 8, n), rng.integers(8, W - w - 8, n)
            cx, cy = rng.integers(1, 4, n), rng.integers(1, 4, n)
            coeff = (cx | (cy << 4)) if name == "hvpp" else cx
            odt = p.dtype if name in ("hpp", "vpp", "vsp", "hvpp") else np.int16
            out = DevBuf.zeros((n, h, w), odt)
            check(L.x265hip_interp_batch(k, 8, depth, w, h, dsrc.ptr, W, out.ptr, w, dev_keep(ys * W + xs), dev_keep(np.arange(n) * w * h),
                                         dev_keep(coeff), 0, n, None))
            got = out.get()
            for i in range(n):
                want = o.interp(name, 0, w, h, src, (int(ys[i]), int(xs[i])), int(cx[i]), int(cy[i]))
                if not np.array_equal(got[i], want):
                    bad.append("%s %dx%d job%d" % (name, w, h, i))
                    break
    _report(bad, 1)


def dev_keep(values):
    from x265_amd.hipprim import dev_i32
    b = dev_i32(values)
    _KEEPALIVE.append(b)
    return b.ptr


# ---- motion estimation ---------------------------------------------------------------------------------------------------
SEA_SHAPES = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (32, 24), (24, 32), (64, 48), (48, 64), (64, 16), (16, 64),
              (16, 12), (12, 16), (16, 4), (4, 16)]      # the shapes whose four sub-blocks lie inside the PU (the others read the reference's stale source cache)


@pytest.mark.parametrize("depth", DEPTHS)
def test_sea_search_matches_oracle(hipmod, depth):
    """--me sea (X265_SEA, motion.cpp:1242-1395): the twelve window-sum planes built on the device equal the oracle's (itself pinned to the
    reference's integral_init* primitives), and the search — DC lower-bound elimination per row against the running best, survivors SAD-ed in
    threes, with the reference's cost arithmetic as it stands — returns the oracle's vector and cost for every PU of a batch."""
    o, g = Orc(depth), hipmod.Hip(depth)
    rng = np.random.default_rng(377 + depth)
    refp, srcp, m = me_scene(depth, 199 + depth)
    H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
    want_planes, got_planes = o.integral_planes(refp), g.integral_planes(refp)
    for k, (w, h) in enumerate(Orc.SEA_WINDOWS):
        assert np.array_equal(want_planes[k], got_planes[k]), (k, w, h)
    bad, total = [], 0
    for qp in (22, 37):
        g.set_mvcost_table(qp, o.mvcost_table(qp))
    for subme in (0, 2, 3):
        for (w, h) in SEA_SHAPES:
            npu = 8
            merange = int(rng.choice([8, 16, 57]))
            qp = int(rng.choice([22, 37]))
            numCand = int(rng.integers(0, 3))
            pus, mins, maxs, mvps, cands = [], [], [], [], []
            for _ in range(npu):
                bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                mr = min(merange, 20)                      # keep every window (incl. the rounding up to 4) inside the 80-pixel margin
                mvmin = ((qmvp[0] >> 2) - mr, (qmvp[1] >> 2) - mr)
                mvmax = ((qmvp[0] >> 2) + mr, (qmvp[1] >> 2) + mr)
                if rng.integers(0, 3) == 0:
                    mvmax = (mvmax[0], min(mvmax[1], int(rng.integers(0, 6))))
                pus.append((bx, by)); mins.append(mvmin); maxs.append(mvmax); mvps.append(qmvp)
                cands.append([(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(numCand)])
            cost, mv = g.motion_estimate_sea_batch(refp, srcp, w, h, pus, mins, maxs, mvps, cands if numCand else [], merange, subme, qp)
            for i in range(npu):
                a = o.motion_estimate_sea(refp, srcp, pus[i][0], pus[i][1], w, h, mins[i], maxs[i], mvps[i], cands[i], merange, subme, qp, planes=want_planes)
                b = (int(cost[i]), (int(mv[i, 0]), int(mv[i, 1])))
                total += 1
                if a != b:
                    bad.append((subme, w, h, pus[i], mvps[i], a, b))
    assert not bad, (len(bad), total, bad[:5])
    assert total >= 400


@pytest.mark.parametrize("planes", [0, 1], ids=["filter", "planes"])
@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [0, 1, 2, 3, 5])
def test_motion_estimate_matches_oracle(hipmod, depth, method, planes):
    o, g = Orc(depth), hipmod.Hip(depth)
    rng = np.random.default_rng(77 + depth + method)
    refp, srcp, m = me_scene(depth, 99 + depth)
    H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 24), (12, 16), (64, 48), (16, 4), (8, 32), (4, 8), (8, 4), (24, 32), (48, 64)]
    bad, total = [], 0
    for qp in (22, 37):
        g.set_mvcost_table(qp, o.mvcost_table(qp))
    for subme in (0, 1, 2, 3, 5, 7):
        for (w, h) in sizes:
            npu = 3 if method == 5 else 12
            merange = 10 if method == 5 else int(rng.choice([16, 32, 57]))
            qp = int(rng.choice([22, 37]))
            numCand = int(rng.integers(0, 4))
            pus, mins, maxs, mvps, cands = [], [], [], [], []
            for _ in range(npu):
                bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                mvmin = ((qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange)
                mvmax = ((qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange)
                if rng.integers(0, 3) == 0:
                    mvmax = (mvmax[0], min(mvmax[1], int(rng.integers(0, 6))))
                pus.append((bx, by)); mins.append(mvmin); maxs.append(mvmax); mvps.append(qmvp)
                cands.append([(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(numCand)])
            cost, mv = g.motion_estimate_batch(refp, srcp, w, h, pus, mins, maxs, mvps, cands if numCand else [], merange, method, subme, qp,
                                               planes_margin=m if planes else 0)
            hipmod._release()
            for i in range(npu):
                want = o.motion_estimate(refp, srcp, pus[i][0], pus[i][1], w, h, mins[i], maxs[i], mvps[i], cands[i], merange, method, subme, qp)
                total += 1
                if (int(cost[i]), (int(mv[i, 0]), int(mv[i, 1]))) != want:
                    bad.append("me%d %dx%d subme%d got %s want %s" % (method, w, h, subme, (int(cost[i]), tuple(int(v) for v in mv[i])), want))
    _report(bad, total)


@pytest.mark.parametrize("planes", [0, 1], ids=["filter", "planes"])
@pytest.mark.parametrize("depth", DEPTHS)
def test_umh_search_matches_oracle_and_golden(hipmod, depth, planes):
    """X265_UMH_SEARCH on the scenes that reach its early-termination, cross and adaptive-range branches (tests/cases.py umh_scenes),
    against the oracle and against the vectors of the real reference committed in tests/golden."""
    import json
    import os
    from cases import umh_groups
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"][str(depth)]["umh"]
    o, g = Orc(depth), hipmod.Hip(depth)
    for qp in (22, 37):
        g.set_mvcost_table(qp, o.mvcost_table(qp))
    bad, total = [], 0
    for gi, (si, refp, srcp, grp) in enumerate(umh_groups(depth)):
        w, h, numCand = grp["w"], grp["h"], grp["numCand"]
        cost, mv = g.motion_estimate_batch(refp, srcp, w, h, grp["pus"], grp["mins"], grp["maxs"], grp["mvps"], grp["cands"] if numCand else [],
                                           grp["merange"], 2, grp["subme"], grp["qp"], planes_margin=96 if planes else 0)
        hipmod._release()
        for i in range(len(grp["pus"])):
            want = o.motion_estimate(refp, srcp, grp["pus"][i][0], grp["pus"][i][1], w, h, grp["mins"][i], grp["maxs"][i], grp["mvps"][i],
                                     grp["cands"][i], grp["merange"], 2, grp["subme"], grp["qp"])
            got = (int(cost[i]), (int(mv[i, 0]), int(mv[i, 1])))
            total += 1
            if got != want or [got[0], got[1][0], got[1][1]] != gold["umh#%d.%d.%d" % (si, gi, i)]:
                bad.append("umh scene %d %dx%d subme%d got %s want %s" % (si, w, h, grp["subme"], got, want))
    _report(bad, total)


@pytest.mark.parametrize("depth", DEPTHS)
def test_misc_batches(hipmod, depth):
    """cpy shuffles, copy_cnt, blockfill, denoise and the RDOQ group costs with many jobs per launch."""
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check
    from oracle import pyoracle as po
    O = po.oracle()
    L = hp.lib()
    rng = np.random.default_rng(600 + depth)
    bad = []
    for size in (4, 8, 16, 32):
        n, S = 211, 700
        nc = size * size
        log2n = size.bit_length() - 1
        plane = short_buf(rng, "rand", (size + 60, S), -2000, 2000)
        plane *= (rng.integers(0, 3, plane.shape) == 0)
        ys, xs = rng.integers(0, 60, n), rng.integers(0, S - size, n)
        dp, doff = DevBuf(plane), _keep(np.asarray(ys * S + xs, np.int32))
        for kind, fn, shift in ((0, O.orc_cpy2Dto1D_shl, 2), (1, O.orc_cpy2Dto1D_shr, 3)):
            d = DevBuf.zeros((n, nc), np.int16)
            check(L.x265hip_cpy_shift_batch(kind, size, d.ptr, dp.ptr, S, doff.ptr, shift, n, None))
            want = np.zeros((n, nc), np.int16)
            for i in range(n):
                fn(po.ptr(want, i, 0), po.ptr(plane, int(ys[i]), int(xs[i])), S, shift, size)
            if not np.array_equal(d.get(), want):
                bad.append("cpy2Dto1D kind%d %d" % (kind, size))
        d, ns = DevBuf.zeros((n, nc), np.int16), DevBuf.zeros((n,), np.uint32)
        check(L.x265hip_copy_cnt_batch(size, d.ptr, dp.ptr, S, doff.ptr, n, ns.ptr, None))
        want, wns = np.zeros((n, nc), np.int16), np.zeros(n, np.uint32)
        for i in range(n):
            wns[i] = O.orc_copy_cnt(po.ptr(want, i, 0), po.ptr(plane, int(ys[i]), int(xs[i])), S, size)
        if not (np.array_equal(d.get(), want) and np.array_equal(ns.get(), wns)):
            bad.append("copy_cnt %d" % size)
        # scatter back (1D -> 2D) on a disjoint grid, and blockfill
        gx, gy = (np.arange(n) % 16) * 40, (np.arange(n) // 16) * 36
        S2 = 16 * 40
        goff = _keep(np.asarray(gy * S2 + gx, np.int32))
        for kind, fn, shift in ((2, O.orc_cpy1Dto2D_shl, 1), (3, O.orc_cpy1Dto2D_shr, 2)):
            out = DevBuf.zeros((int(gy.max()) + 36, S2), np.int16)
            check(L.x265hip_cpy_shift_batch(kind, size, out.ptr, _keep(want).ptr, S2, goff.ptr, shift, n, None))
            w2 = np.zeros(out.shape, np.int16)
            for i in range(n):
                fn(po.ptr(w2, int(gy[i]), int(gx[i])), po.ptr(want, i, 0), S2, shift, size)
            if not np.array_equal(out.get(), w2):
                bad.append("cpy1Dto2D kind%d %d" % (kind, size))
        vals = rng.integers(-300, 300, n).astype(np.int16)
        out = DevBuf.zeros((int(gy.max()) + 36, S2), np.int16)
        check(L.x265hip_blockfill_s_batch(size, out.ptr, S2, goff.ptr, _keep(vals).ptr, n, None))
        w2 = np.zeros(out.shape, np.int16)
        for i in range(n):
            w2[gy[i]:gy[i] + size, gx[i]:gx[i] + size] = vals[i]
        if not np.array_equal(out.get(), w2):
            bad.append("blockfill %d" % size)
        # RDOQ costs: every coefficient group of every TU, psy and non-psy
        resi = short_buf(rng, "rand", (n, nc), -32768, 32767)
        fenc = short_buf(rng, "rand", (n, nc), -32768, 32767)
        cgs = [(t, cy * 4 * size + cx * 4) for t in range(n) for cy in range(size // 4) for cx in range(size // 4)]
        tu = _keep(np.array([c[0] for c in cgs], np.int32))
        bp = _keep(np.array([c[1] for c in cgs], np.int32))
        psy = np.array([int(rng.integers(1, 1 << 20))], np.int64)
        for kind, name in ((0, "orc_nonpsy_rdoquant"), (1, "orc_psy_rdoquant")):
            cu = DevBuf.zeros((n, nc), np.int64)
            a, b = DevBuf.zeros((len(cgs),), np.int64), DevBuf.zeros((len(cgs),), np.int64)
            check(L.x265hip_rdoq_cost_batch(kind, size, depth, _keep(resi).ptr, _keep(fenc).ptr, _keep(psy).ptr, tu.ptr, bp.ptr, len(cgs),
                                            cu.ptr, a.ptr, b.ptr, None))
            wcu = np.zeros((n, nc), np.int64)
            wa = np.zeros(len(cgs), np.int64)
            m = min(400, len(cgs))
            for j, (t, p) in enumerate(cgs[:m]):
                tot = np.zeros(2, np.int64)
                if kind == 0:
                    O.orc_nonpsy_rdoquant(log2n, po.ptr(resi, t, 0), po.ptr(wcu, t, 0), po.vp(tot.ctypes.data), po.vp(tot.ctypes.data + 8), p, depth)
                else:
                    O.orc_psy_rdoquant(log2n, po.ptr(resi, t, 0), po.ptr(fenc, t, 0), po.ptr(wcu, t, 0), po.vp(tot.ctypes.data),
                                       po.vp(tot.ctypes.data + 8), po.ptr(psy), p, depth)
                wa[j] = tot[0]
            gcu, ga = cu.get(), a.get()
            tmax = cgs[m - 1][0]
            if not (np.array_equal(ga[:m], wa[:m]) and np.array_equal(gcu[:tmax], wcu[:tmax]) and np.array_equal(ga, b.get())):
                bad.append("rdoq kind%d %d" % (kind, size))
    _report(bad, 1)


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_intra_pred_batch_all_modes(hipmod, depth):
    """Every mode x bFilter of every size in ONE launch per size, lines from the TestBench distribution + extremes."""
    o, g = Orc(depth), hipmod.Hip(depth)
    rng = np.random.default_rng(41 + depth)
    pmax = (1 << depth) - 1
    for n in (4, 8, 16, 32):
        lines, modes, bfs = [], [], []
        for trial in range(6):
            if trial == 0:
                ln = np.full(4 * n + 1, pmax)
            elif trial == 1:
                ln = np.zeros(4 * n + 1)
            else:
                ln = rng.integers(0, pmax + 1, 4 * n + 1)
            for m in range(35):
                for bf in (0, 1):
                    lines.append(ln.astype(o.pix)); modes.append(m); bfs.append(bf)
        lines = np.ascontiguousarray(np.stack(lines))
        got = g.intra_pred_batch(n, lines, modes, bfs)
        hipmod._release()
        for i in range(len(modes)):
            assert np.array_equal(got[i], o.intra_pred(n, modes[i], lines[i], bfs[i])), (n, modes[i], bfs[i])


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_intra_mode_scan_matches_oracle(hipmod, depth):
    """x265hip_intra_scan_batch = for every block and each of the 35 modes: intra_pred (raw / filtered line and bFilter as
    Search::checkIntraInInter chooses them) followed by cu[].sa8d against the source block — composed here from the two pinned
    oracle primitives."""
    from cases import textured_frame
    o, g = Orc(depth), hipmod.Hip(depth)
    rng = np.random.default_rng(61 + depth)
    plane = textured_frame(rng, 160, 224, depth)
    for n in (4, 8, 16, 32):
        count = 23
        xy = [(int(rng.integers(1, 160 - n)), int(rng.integers(1, 224 - 2 * n))) for _ in range(count)]
        lines = np.zeros((count, 4 * n + 1), o.pix)
        for i, (y, x) in enumerate(xy):
            if i % 5 == 4:
                lines[i] = rng.integers(0, o.pmax + 1, 4 * n + 1)            # unrelated neighbours: large costs, clipping edge gradients
            else:
                lines[i, 0] = plane[y - 1, x - 1]
                lines[i, 1:2 * n + 1] = np.resize(plane[y - 1, x:x + 2 * n], 2 * n)
                lines[i, 2 * n + 1:] = np.resize(plane[y:y + 2 * n, x - 1], 2 * n)
        filt = np.stack([o.intra_filter(n, lines[i]) for i in range(count)])
        got = g.intra_scan(n, lines, filt, plane, xy)
        hipmod._release()
        for i, (y, x) in enumerate(xy):
            for mode in range(35):
                src = filt[i] if o.intra_uses_filtered(n, mode) else lines[i]
                pred = o.intra_pred(n, mode, src, 1 if n <= 16 else 0)
                want = o.sa8d(n, plane, (y, x), pred, (0, 0)) if n > 4 else o.satd(4, 4, plane, (y, x), pred, (0, 0))
                assert got[i, mode] == want, (n, i, mode, int(got[i, mode]), want)


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_lowres_pass_matches_oracle(hipmod, depth):
    """Lowres::init (downscale + four extended planes) and the lookahead intra estimate: planes, per-block cost and mode,
    row sums and frame estimate, bit for bit; odd sizes exercise the rounded-up lowres width and dead rows of the last team."""
    from cases import lowres_scene
    o, g = Orc(depth), hipmod.Hip(depth)
    for i, (w, h) in enumerate([(200, 136), (176, 144), (66, 50), (1920, 1080)]):
        src, m = lowres_scene(depth, 500 + depth + i, h, w)
        a = o.lowres_pass(src, (m, m), w, h, m, m)
        b = g.lowres_pass(src, (m, m), w, h, m, m)
        hipmod._release()
        lw = a[5][1]
        assert a[5] == b[5]
        assert a[0] == b[0], (w, h, a[0], b[0])
        for k in (1, 2, 3):
            assert np.array_equal(a[k], b[k]), (w, h, k, np.argwhere(a[k] != b[k])[:5])
        for pa, pb in zip(a[4], b[4]):
            assert np.array_equal(pa[:, :lw + 2 * m], pb[:, :lw + 2 * m]), (w, h)


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_chroma_motion_estimate_matches_oracle(hipmod, depth):
    """motionEstimate with the chroma SATD term of subpelCompare (subme > 2, 4:2:0): every PU shape x DIA / HEX / STAR x subme
    2..7 one by one, then a frame-shaped batch of 16x16 PUs at subme 3 (BASELINE configs[2])."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from cases import me_scene_yuv
    want = make_golden.chroma_me_results(Orc, depth)
    got = make_golden.chroma_me_results(hipmod.Hip, depth)
    hipmod._release()
    bad = [k for k in want if want[k] != got[k]]
    assert not bad, (len(bad), bad[:8], [(want[k], got[k]) for k in bad[:4]])
    o, g = Orc(depth), hipmod.Hip(depth)
    ref, src, m = me_scene_yuv(depth, 321 + depth)
    pus = [(m + x, m + y) for y in range(0, 160, 16) for x in range(0, 192, 16)]
    qmvp = [((i * 7) % 23 - 11, (i * 5) % 19 - 9) for i in range(len(pus))]
    mvmin = [((q[0] >> 2) - 16, (q[1] >> 2) - 16) for q in qmvp]
    mvmax = [((q[0] >> 2) + 16, (q[1] >> 2) + 16) for q in qmvp]
    g.motion_estimate_chroma(ref, src, pus[0][0], pus[0][1], 16, 16, mvmin[0], mvmax[0], qmvp[0], [], 16, 3, 3, 28)   # loads the cost table
    cost, mv = g.motion_estimate_chroma_batch(ref, src, 16, 16, pus, mvmin, mvmax, qmvp, [], 16, 3, 3, 28)
    hipmod._release()
    for i, (bx, by) in enumerate(pus):
        c, v = o.motion_estimate_chroma(ref, src, bx, by, 16, 16, mvmin[i], mvmax[i], qmvp[i], [], 16, 3, 3, 28)
        assert (int(cost[i]), int(mv[i, 0]), int(mv[i, 1])) == (c, v[0], v[1]), i


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_bipred_matches_oracle(hipmod, depth):
    """Bi-predictive motion compensation (two 14-bit predictions + addAvg; luma, Cb, Cr): every PU shape one by one (the golden case
    list), then a frame-shaped batch of 16x16 PUs with per-PU vector pairs, nothing written outside the PUs."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from cases import me_scene_yuv
    want = make_golden.bipred_results(Orc, depth)
    got = make_golden.bipred_results(hipmod.Hip, depth)
    hipmod._release()
    bad = [k for k in want if not same(want[k], got[k])]
    assert not bad, (len(bad), bad[:8])
    o, g = Orc(depth), hipmod.Hip(depth)
    ref, src, m = me_scene_yuv(depth, 55 + depth)
    rng = np.random.default_rng(12)
    pus = [(m + x, m + y) for y in range(0, 160, 16) for x in range(0, 192, 16)]
    mv0 = [(int(rng.integers(-30, 31)), int(rng.integers(-30, 31))) for _ in pus]
    mv1 = [(int(rng.integers(-30, 31)), int(rng.integers(-30, 31))) for _ in pus]
    y, cb, cr = g.pred_inter_bi_batch(ref, src, 16, 16, pus, mv0, mv1)
    hipmod._release()
    wy, wcb, wcr = np.zeros_like(y), np.zeros_like(cb), np.zeros_like(cr)
    for (bx, by), a, b in zip(pus, mv0, mv1):
        py, pcb, pcr = o.pred_inter_bi(ref, src, bx, by, 16, 16, a, b)
        wy[by:by + 16, bx:bx + 16] = py
        wcb[by // 2:by // 2 + 8, bx // 2:bx // 2 + 8] = pcb
        wcr[by // 2:by // 2 + 8, bx // 2:bx // 2 + 8] = pcr
    assert np.array_equal(y, wy) and np.array_equal(cb, wcb) and np.array_equal(cr, wcr)


@pytest.mark.parametrize("depth", DEPTHS)
def test_loop_filter_primitives_match_oracle_and_golden(hipmod, depth):
    """Deblocking edge filters, SAO offset application and SAO statistics: every case against the oracle and the committed digests of the
    reference; then whole-picture launches — every vertical luma edge segment of a picture in one launch, band offset and statistics of
    every CTU in one launch — against per-call oracle results."""
    import json
    import itertools
    from cases import loop_cases, deblock_cases, digest
    from x265_amd.hipprim import DevBuf, check, dev_i32, SaoJob, SaoStatsJob
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"][str(depth)]["loop"]
    o, g = Orc(depth), hipmod.Hip(depth)
    bad, total = [], 0
    for label, fn, args in itertools.chain(loop_cases(depth), deblock_cases(depth)):
        want, got = getattr(o, fn)(*args), getattr(g, fn)(*args)
        hipmod._release()
        total += 1
        if not same(want, got) or digest(got) != gold[label]:
            bad.append(label)
    _report(bad, total)
    rng = np.random.default_rng(17 + depth)
    H, W = 136, 200
    pic = (np.kron(rng.integers(0, (1 << depth) - 8, size=(H // 4, W // 4)), np.ones((4, 4), np.int64)) + rng.integers(0, 3, size=(H, W))).astype(g.pix)
    L = g.L
    # all vertical edges on the 8-sample grid, four lines each: disjoint jobs, one launch
    segs = [(y, x) for y in range(0, H, 4) for x in range(8, W, 8)]
    tcP, tcQ = rng.integers(0, 20, size=len(segs)).astype(np.int32), rng.integers(0, 20, size=len(segs)).astype(np.int32)
    d, doff, dp, dq = DevBuf(pic), DevBuf(np.array([y * W + x for (y, x) in segs], np.int64)), dev_i32(tcP), dev_i32(tcQ)   # named: launches are asynchronous
    check(L.x265hip_pel_filter_luma_strong_batch(depth, d.ptr, doff.ptr, W, 1, dp.ptr, dq.ptr, len(segs), None))
    want = pic
    for i, (y, x) in enumerate(segs):
        want = o.pel_filter_luma_strong(want, (y, x), 0, int(tcP[i]), int(tcQ[i]))
    assert np.array_equal(d.get(), want)
    # the deblocking of a whole picture in two launches: every unit of every vertical edge of the 8-sample grid, then every horizontal one,
    # with per-unit boundary strengths and QPs — against the per-unit restatement applied in the same order
    from oracle import pyoracle as po
    fl = o._lf("orc_deblock_luma_unit", [po.vp, po.ip, po.ip] + [po.i32] * 9)
    blocky = (np.kron(np.cumsum(rng.integers(-6, 7, size=(H // 8, W // 8)), axis=1) + 120, np.ones((8, 8), np.int64)) << (depth - 8)).astype(g.pix)
    blocky = np.clip(blocky.astype(np.int64) + rng.integers(-1, 2, size=blocky.shape), 0, (1 << depth) - 1).astype(g.pix)
    d, want = DevBuf(blocky), blocky.copy()
    for edgeDir in (0, 1):
        units = [(x, y) for y in range(0, H, 4) for x in range(8, W, 8)] if edgeDir == 0 else [(x, y) for y in range(8, H, 8) for x in range(0, W, 4)]
        bs = rng.integers(0, 3, size=len(units)).astype(np.uint8)
        qpP, qpQ = rng.integers(18, 46, size=len(units)).astype(np.int8), rng.integers(18, 46, size=len(units)).astype(np.int8)
        dxy, dbs, dp_, dq_ = dev_i32(np.array(units, np.int32).reshape(-1)), DevBuf(bs), DevBuf(qpP), DevBuf(qpQ)
        check(L.x265hip_deblock_luma_batch(depth, d.ptr, W, edgeDir, dxy.ptr, dbs.ptr, dp_.ptr, dq_.ptr, None, 1, -1, len(units), None))
        step, off = (W, 1) if edgeDir == 0 else (1, W)
        for i, (x, y) in enumerate(units):
            fl(o_ptr(want, y, x), step, off, int(bs[i]), int(qpP[i]), int(qpQ[i]), 0, 0, 0, 1, -1, depth)
        got = d.get()
        assert np.array_equal(got, want), (edgeDir, int((got != want).sum()))
    assert not np.array_equal(want, blocky)
    # band offset and BO / E0 statistics of every 64x64 CTU (ragged right and bottom ones) in one launch each
    ctus = [(y, x, min(64, W - 1 - x), min(64, H - 1 - y)) for y in range(1, H - 1, 64) for x in range(1, W - 1, 64)]
    jobs = (SaoJob * len(ctus))()
    band = rng.integers(-7, 8, size=(len(ctus), 32)).astype(np.int8)
    for i, (y, x, w, h) in enumerate(ctus):
        jobs[i].recOff, jobs[i].width, jobs[i].height = y * W + x, w, h
        for k in range(32):
            jobs[i].offsets[k] = int(band[i, k])
    d, dj = DevBuf(pic), DevBuf(np.frombuffer(bytes(jobs), np.uint8).copy())
    check(L.x265hip_sao_apply_batch(depth, 5, d.ptr, W, None, dj.ptr, len(ctus), None))
    want = pic
    for i, (y, x, w, h) in enumerate(ctus):
        want = o.sao_b0(want, (y, x), band[i], w, h)
    assert np.array_equal(d.get(), want)
    diff = rng.integers(-200, 201, size=(len(ctus), 64, 64)).astype(np.int16)
    for kind in (0, 1):
        sj = (SaoStatsJob * len(ctus))()
        for i, (y, x, w, h) in enumerate(ctus):
            sj[i].diffOff, sj[i].recOff, sj[i].endX, sj[i].endY = i * 4096, y * W + x, w, h
        ds, dc = DevBuf.zeros((len(ctus), 32), np.int32), DevBuf.zeros((len(ctus), 32), np.int32)
        dd, dpic, dsj = DevBuf(diff), DevBuf(pic), DevBuf(np.frombuffer(bytes(sj), np.uint8).copy())
        check(L.x265hip_sao_stats_batch(depth, kind, dd.ptr, dpic.ptr, W, None, dsj.ptr, len(ctus), ds.ptr, dc.ptr, None))
        st, ct = ds.get(), dc.get()
        ncls = 32 if kind == 0 else 5
        for i, (y, x, w, h) in enumerate(ctus):
            ws, wc, _, _ = o.sao_stats(kind, diff[i], pic, (y, x), w, h, np.zeros(ncls, np.int32), np.zeros(ncls, np.int32), np.zeros(w + 2, np.int8),
                                       np.zeros(w + 2, np.int8))
            assert st[i][:ncls].tolist() == ws.tolist() and ct[i][:ncls].tolist() == wc.tolist(), (kind, i)
    hipmod._release()


def test_coefficient_scan_primitives_match_oracle_and_golden(hipmod):
    """scanPosLast / findPosFirstLast / costCoeffNxN / costCoeffRemain / costC1C2Flag: every case one by one against the oracle and the
    committed digests of the reference, then many jobs per launch (a launch of 300 TUs per size, 256-job cost batches with private
    context copies) against per-job oracle results."""
    import json
    from cases import coef_cases, digest
    from backends import scan_order_py
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"]["coef"]
    o, g = Orc(8), hipmod.Hip(8)
    bad, total = [], 0
    for label, fn, args in coef_cases():
        want, got = getattr(o, fn)(*args), getattr(g, fn)(*args)
        total += 1
        if not same(want, got) or digest(got) != gold[label]:
            bad.append(label)
    hipmod._release()
    _report(bad, total)
    rng = np.random.default_rng(5)
    for log2 in (2, 3, 4, 5):
        size = 1 << log2
        for stype in range(3):
            tus = rng.integers(-300, 301, size=(300, size, size)).astype(np.int16)
            tus[rng.random(tus.shape) < 0.8] = 0
            tus[:, -1, -1] |= 1                                  # at least one coefficient each
            tus[7] = 0
            tus[7, 0, 0] = -5
            last, sign, flag, num = g.scan_pos_last_batch(log2, stype, tus)
            for i in range(0, 300, 7):
                w = o.scan_pos_last(log2, stype, tus[i])
                assert (int(last[i]), sign[i].tolist(), flag[i].tolist(), num[i].tolist()) == (w[0], w[1].tolist(), w[2].tolist(), w[3].tolist()), (log2, stype, i)
            tu = tus[3]
            cgs = [(x, y) for y in range(size // 4) for x in range(size // 4) if tu[y * 4:y * 4 + 4, x * 4:x * 4 + 4].any()]
            if cgs:
                got = g.find_pos_first_last_batch(tu, cgs, stype)
                assert [int(v) for v in got] == [o.find_pos_first_last(tu, x, y, stype) for (x, y) in cgs]
            ncg = (size // 4) ** 2
            jobs = [(int(rng.integers(0, ncg)), int(rng.integers(0, 16)), int(rng.integers(0, 4)), 0 if log2 == 2 else (9 if log2 == 3 else 12),
                     rng.integers(2, 125, size=64).astype(np.uint8)) for _ in range(256)]
            bits, absC, ctxs = g.cost_coeff_nxn_batch(tu, log2, stype, jobs)
            for i in range(0, 256, 5):
                w = o.cost_coeff_nxn(tu, log2, stype, *jobs[i])
                first = 1 if jobs[i][1] < 15 else 0             # the oracle wrapper's buffer starts `first` slots earlier (see hipbackend)
                assert (int(bits[i]), absC[i][:16 - first].tolist(), ctxs[i].tolist()) == (w[0], w[1][first:].tolist(), w[2].tolist()), (log2, stype, i)
    hipmod._release()
    a = rng.integers(1, 200, size=(512, 16)).astype(np.uint16)
    a[rng.random(a.shape) < 0.5] = 1
    nnz = rng.integers(1, 17, size=512)
    idx = np.array([int(rng.integers(0, n)) for n in nnz])
    got = g.cost_coeff_remain_batch(a, nnz, idx)
    assert [int(v) for v in got] == [o.cost_coeff_remain(a[i], int(nnz[i]), int(idx[i])) for i in range(512)]
    cnt = rng.integers(1, 9, size=512)
    ctx = rng.integers(2, 125, size=(512, 8)).astype(np.uint8)
    a = rng.integers(1, 5, size=(512, 16)).astype(np.uint16)
    out, c2 = g.cost_c1c2_flag_batch(a, cnt, ctx, 5)
    for i in range(512):
        w = o.cost_c1c2_flag(a[i], int(cnt[i]), ctx[i], 5)
        assert (int(out[i]), c2[i][:8].tolist()) == (w[0], w[1].tolist()), i


@pytest.mark.parametrize("depth", DEPTHS)
def test_motion_compensation_matches_oracle_and_golden(hipmod, depth):
    """Predict::motionCompensation, every branch (P / B-uni / bi x weighted prediction off, on-but-absent, present) for every PU shape
    against the oracle and the committed vectors of the real reference; then a frame-shaped weighted bi-predictive batch."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from cases import me_scene_yuv, digest
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"][str(depth)]["mc"]
    want = make_golden.mc_results(Orc, depth)
    got = make_golden.mc_results(hipmod.Hip, depth)
    hipmod._release()
    bad = [k for k in want if not same(want[k], got[k]) or digest(got[k]) != gold[k]]
    assert len(want) >= 240 and not bad, (len(bad), bad[:8])
    o, g = Orc(depth), hipmod.Hip(depth)
    ref, src, m = me_scene_yuv(depth, 58 + depth)
    rng = np.random.default_rng(13)
    pus = [(m + x, m + y) for y in range(0, 160, 16) for x in range(0, 192, 16)]
    mv0 = [(int(rng.integers(-30, 31)), int(rng.integers(-30, 31))) for _ in pus]
    mv1 = [(int(rng.integers(-30, 31)), int(rng.integers(-30, 31))) for _ in pus]
    wp0, wp1 = [(70, 5, 6, 1), (60, -3, 6, 1), (33, 2, 5, 0)], [(61, -7, 6, 0), (64, 0, 6, 0), (29, 1, 5, 1)]
    for r1, v1, w1 in ((src, mv1, wp1), (None, None, None)):
        y, cb, cr = g.motion_compensation_batch(ref, r1, 16, 16, pus, mv0, v1, wp0, w1)
        hipmod._release()
        wy, wcb, wcr = np.zeros_like(y), np.zeros_like(cb), np.zeros_like(cr)
        for i, (bx, by) in enumerate(pus):
            py, pcb, pcr = o.motion_compensation(ref, r1, bx, by, 16, 16, mv0[i], v1[i] if v1 else None, wp0, w1)
            wy[by:by + 16, bx:bx + 16] = py
            wcb[by // 2:by // 2 + 8, bx // 2:bx // 2 + 8] = pcb
            wcr[by // 2:by // 2 + 8, bx // 2:bx // 2 + 8] = pcr
        assert np.array_equal(y, wy) and np.array_equal(cb, wcb) and np.array_equal(cr, wcr)


def test_cutree_propagate_matches_oracle_and_golden(hipmod):
    """Lookahead::estimateCUPropagate on the device (double-precision amounts, scatter into the references with saturation) vs the restatement
    and the committed results of the real function, up to the block grid of a 1080p frame."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from cases import digest
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"]["cutree"]
    want = make_golden.cutree_results(Orc)
    got = make_golden.cutree_results(hipmod.Hip)
    hipmod._release()
    for k in want:
        assert np.array_equal(want[k][0], got[k][0]) and np.array_equal(want[k][1], got[k][1]), k
        assert [digest(got[k][0]), digest(got[k][1])] == gold[k][:2], k


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_weightp_analysis_matches_oracle_and_golden(hipmod, depth):
    """The lookahead's weighted-prediction analysis on the device (lowres planes, intra costs, the two weightCostLuma evaluations, the
    decision, the weighted planes), the P-frame cost pass on the weighted planes and the adaptive-quantisation frame pass
    (calcAdaptiveQuantFrame) vs the restatement and (8 / 10 bit) the committed results of the real weightsAnalyse; then a batch of
    candidate weights in one launch against per-candidate oracle costs."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from cases import weight_scenes, digest
    from x265_amd.hipprim import DevBuf, check, WeightParam
    want = make_golden.weightp_results(Orc, depth)
    got = make_golden.weightp_results(hipmod.Hip, depth)
    hipmod._release()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"].get(str(depth), {}).get("weightp")
    for k in want:
        assert same(want[k], got[k]), k
        if gold:
            assert digest(got[k]) == gold[k], k
    assert sum(v[0] for v in want.values()) >= 3
    # the P-frame cost pass with --weightp, everything on the device: analysis, then the pass with the weighted planes as list 0
    want = make_golden.lookahead_weightp_results(Orc, depth)
    got = make_golden.lookahead_weightp_results(hipmod.Hip, depth)
    hipmod._release()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"].get(str(depth), {}).get("lookahead_weightp")
    for k in want:
        assert len(want[k]) == len(got[k]) and all(same(x, y) for x, y in zip(want[k], got[k])), k
        if gold:
            assert digest(got[k]) == gold[k], k
    # adaptive quantisation (energies on the device, the offsets in double precision on the host side of the library)
    want = make_golden.aq_results(Orc, depth)
    got = make_golden.aq_results(hipmod.Hip, depth)
    hipmod._release()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_golden.json")))["golden"].get(str(depth), {}).get("aq")
    for k in want:
        assert want[k][0] == got[k][0] and all(np.array_equal(x, y) for x, y in zip(want[k][1:], got[k][1:])), k
        if gold:
            assert digest(got[k]) == gold[k], k
    o, g = Orc(depth), hipmod.Hip(depth)
    label, s0, s1, m, H, W, st = next(iter(weight_scenes(depth)))
    _, icost, _, _, pl1, (stride, lw, lh) = o.lowres_pass(s1, (m, m), W, H, m, m)
    _, _, _, _, pl0, _ = o.lowres_pass(s0, (m, m), W, H, m, m)
    rng = np.random.default_rng(3)
    cands = [(int(rng.integers(1, 128)), int(rng.integers(-60, 61)), int(rng.integers(0, 8)), int(rng.integers(0, 2))) for _ in range(24)]
    arr = (WeightParam * len(cands))(*[WeightParam(*c) for c in cands])
    df, dr, di, dc = DevBuf(pl1[0]), DevBuf(pl0[0]), DevBuf(icost), DevBuf.zeros((len(cands),), np.uint32)
    org = m * stride + m
    check(g.L.x265hip_lookahead_weight_cost_batch(depth, df.at(org), dr.at(org), stride, lw, lh, di.ptr, arr, len(cands), dc.ptr, None))
    costs = dc.get()
    import ctypes as C
    from oracle import pyoracle as po
    fn = o._f("orc_weight_cost_luma")
    fn.restype = C.c_uint32
    fn.argtypes = [po.vp, po.vp, po.ip, po.i32, po.i32, po.vp, po.i32, po.i32, po.i32, po.i32, po.i32]
    for i, (w_, o_, d_, p_) in enumerate(cands):
        assert int(costs[i]) == fn(o_ptr(pl1[0], m, m), o_ptr(pl0[0], m, m), stride, lw, lh, o_ptr(icost, 0, 0), p_, w_, o_, d_, depth), (i, cands[i])


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_lookahead_p_cost_matches_oracle(hipmod, depth):
    """The lookahead's P-frame cost pass (lowres init -> intra estimate -> estimateCUCost over the frame) on the GPU vs the
    restatement: every block's vector, cost, packed lowresCost, the row sums, the frame score and the intra count; serial and
    sliced; 11 pairs in one launch (not a multiple of the 8 XCDs), run twice on the same handshake scratch."""
    from cases import lookahead_scene
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    o, g = Orc(depth), hipmod.Hip(depth)
    for (w, h, rps, ns) in make_golden.LOOKAHEAD_CASES:
        s0, s1, m = lookahead_scene(depth, 700 + depth + w, h, w)
        a = o.lookahead_cost_p(s0, s1, (m, m), w, h, m, m, rps, ns)
        b = g.lookahead_cost_p(s0, s1, (m, m), w, h, m, m, rps, ns)
        hipmod._release()
        for k, (x, y) in enumerate(zip(a, b)):
            assert same(x, y), (w, h, rps, ns, k)
    w, h = 320, 200
    scenes = [lookahead_scene(depth, 900 + depth + i, h, w) for i in range(11)]
    m = scenes[0][2]
    got = g.lookahead_cost_p_batch([(s[0], s[1]) for s in scenes], (m, m), w, h, m, m, 4, 3)
    hipmod._release()
    for i, sc in enumerate(scenes):
        want = o.lookahead_cost_p(sc[0], sc[1], (m, m), w, h, m, m, 4, 3)
        for k, (x, y) in enumerate(zip(want, got[i])):
            assert same(x, y), (i, k)


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_lookahead_b_cost_matches_oracle(hipmod, depth):
    """B-frame cost pass: both list searches with the skip rule in one launch (or list 0 reused from the P estimate), then the
    bidir / co-located candidates — vectors and costs of both lists, packed lowresCosts, row sums, frame score."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    want = make_golden.lookahead_b_results(Orc, depth)
    got = make_golden.lookahead_b_results(hipmod.Hip, depth)
    hipmod._release()
    for k in want:
        for j, (x, y) in enumerate(zip(want[k], got[k])):
            assert same(x, y), (k, j)


def test_twelve_bit_primitives_match_oracle(hipmod):
    """depth 12 (u16 pixels, the third X265_DEPTH): same sweep against the oracle restatement, whose 12-bit arithmetic is pinned to a Main12
    build of the reference on the CPU side (oracle/Makefile ref12, tests/test_oracle_vs_ref.py DEPTHS)."""
    o, g = Orc(12), hipmod.Hip(12)
    bad, n = [], 0
    for label, fn, args in gen_cases(12, seed=777, reps=1):
        if fn in NOT_ON_GPU:
            continue
        want = getattr(o, fn)(*args)
        got = getattr(g, fn)(*args)
        hipmod._release()
        n += 1
        if not same(got, want):
            bad.append(label)
    assert n > 1500
    _report(bad, n)


def test_empty_batches_and_bad_arguments(hipmod):
    """n = 0 is a no-op that succeeds; impossible shapes / depths are rejected with X265HIP_EINVAL and a message."""
    from x265_amd import hipprim as hp
    L = hp.lib()
    assert L.x265hip_pixcmp_batch(hp.CMP_SAD, 8, 8, 8, None, 0, None, 0, None, None, 0, None, None) == 0
    assert L.x265hip_dct_batch(32, 0, 8, None, 32, None, None, 0, None) == 0
    assert L.x265hip_quant_batch(None, None, None, None, 20, 1, 64, 0, None, None) == 0
    assert L.x265hip_motion_estimate_batch(8, 16, 16, None, 0, None, 0, None, None, None, None, 0, None, 57, 1, 2, None, 65536, 0, None, None, None) == 0
    assert L.x265hip_residual_chain_batch(32, 8, None, 0, None, 0, None, 0, None, None, None, None, 20, 1, 40, 1, None, None, None, 0, None) == 0
    assert L.x265hip_pixcmp_batch(hp.CMP_SAD, 9, 8, 8, None, 0, None, 0, None, None, 1, None, None) == -1      # depth 9
    assert b"depth" in L.x265hip_last_error()
    assert L.x265hip_pixcmp_batch(hp.CMP_SA8D, 8, 16, 8, None, 0, None, 0, None, None, 1, None, None) == -1    # sa8d is square
    assert L.x265hip_dct_batch(64, 0, 8, None, 64, None, None, 1, None) == -1                                  # no 64-point transform
    assert L.x265hip_dct_batch(8, 1, 8, None, 8, None, None, 1, None) == -1                                    # DST is 4x4 only
    assert L.x265hip_motion_estimate_batch(8, 4, 4, None, 0, None, 0, None, None, None, None, 0, None, 57, 1, 2, None, 65536, 1, None, None, None) == -1
    assert L.x265hip_motion_estimate_batch(8, 8, 8, None, 0, None, 0, None, None, None, None, 0, None, 57, 4, 2, None, 65536, 1, None, None, None) == -1   # SEA (needs the integral planes) is not implemented
    assert L.x265hip_interp_batch(hp.IF_HVPP, 4, 8, 8, 8, None, 0, None, 0, None, None, None, 0, 1, None) == -1          # hv_pp is a luma slot
    from x265_amd.framepass import FramePass
    with pytest.raises(hp.HipError):
        FramePass(1921, 1080)                                                                                  # not a multiple of 8
