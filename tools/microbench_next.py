"""Frame-sized timings of the SURVEY.md §8f "next" rows' entry points on one GPU (1920x1080 8-bit shapes, HIP-event timing): the
deblocking of every edge unit of a picture, SAO statistics and band offsets of every CTU, the coefficient-scan helpers over a frame's
transform units, weighted motion compensation, the AQ block energies, candidate weight costs on the lowres planes, the CU-tree step.
Algorithmic bytes: what one call of the reference's function reads + writes, times the calls in the launch.  Not the contract bench."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import hipprim as hp                                           # noqa: E402
from x265_amd.hipprim import DevBuf, check, dev_i32, SaoJob, SaoStatsJob, WeightParam, CoeffGroupJob     # noqa: E402
from x265_amd.framepass import YuvStruct                                     # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    t = hp.Timer(None)
    t.start()
    for _ in range(iters):
        fn()
    return t.stop_ms() / iters


def main():
    L = hp.lib()
    check(L.x265hip_init(0))
    depth, W, H, M = 8, 1920, 1080, 96
    S = W + 2 * M
    rng = np.random.default_rng(2)
    pic = (np.kron(rng.integers(20, 230, size=((H + 2 * M) // 8, S // 8)), np.ones((8, 8), np.int64)) + rng.integers(-2, 3, size=(H + 2 * M, S))).astype(np.uint8)
    org = M * S + M
    res = []

    def row(name, n, ms, byts):
        res.append({"kernel": name, "n": int(n), "us": round(ms * 1000, 2), "GBps": round(byts / ms / 1e6, 1)})

    # deblocking: every unit of every vertical edge of the 8-sample grid, then every horizontal one
    d = DevBuf(pic)
    for edgeDir, units in ((0, [(M + x, M + y) for y in range(0, H, 4) for x in range(8, W, 8)]), (1, [(M + x, M + y) for y in range(8, H, 8) for x in range(0, W, 4)])):
        n = len(units)
        dxy = dev_i32(np.array(units, np.int32).reshape(-1))
        bs, qp = DevBuf(rng.integers(0, 3, size=n).astype(np.uint8)), DevBuf(rng.integers(22, 40, size=n).astype(np.int8))
        ms = timeit(lambda: check(L.x265hip_deblock_luma_batch(depth, d.ptr, S, edgeDir, dxy.ptr, bs.ptr, qp.ptr, qp.ptr, None, 0, 0, n, None)))
        row("deblock luma %s edges" % ("vertical" if edgeDir == 0 else "horizontal"), n, ms, n * (32 + 24 + 4))
    # SAO: band offset and E0 / E2 statistics of every CTU
    ctus = [(y, x, min(64, W - x), min(64, H - y)) for y in range(0, H, 64) for x in range(0, W, 64)]
    jobs = (SaoJob * len(ctus))()
    for i, (y, x, w, h) in enumerate(ctus):
        jobs[i].recOff, jobs[i].width, jobs[i].height = org + y * S + x, w, h
    dj = DevBuf(np.frombuffer(bytes(jobs), np.uint8).copy())
    ms = timeit(lambda: check(L.x265hip_sao_apply_batch(depth, 5, d.ptr, S, None, dj.ptr, len(ctus), None)))
    row("sao band offset, all CTUs", len(ctus), ms, 2 * W * H)
    diff = DevBuf(rng.integers(-50, 51, size=(len(ctus), 64, 64)).astype(np.int16))
    sj = (SaoStatsJob * len(ctus))()
    aux = DevBuf.zeros((len(ctus) * 160,), np.int8)
    for i, (y, x, w, h) in enumerate(ctus):
        sj[i].diffOff, sj[i].recOff, sj[i].endX, sj[i].endY, sj[i].aux0, sj[i].aux1 = i * 4096, org + y * S + x, min(w, 63), min(h, 63), i * 160 + 1, i * 160 + 81
    dsj = DevBuf(np.frombuffer(bytes(sj), np.uint8).copy())
    st, ct = DevBuf.zeros((len(ctus), 32), np.int32), DevBuf.zeros((len(ctus), 32), np.int32)
    for kind, nm in ((0, "BO"), (1, "E0"), (3, "E2")):
        ms = timeit(lambda: check(L.x265hip_sao_stats_batch(depth, kind, diff.ptr, d.ptr, S, aux.ptr, dsj.ptr, len(ctus), st.ptr, ct.ptr, None)))
        row("sao stats %s, all CTUs" % nm, len(ctus), ms, W * H * 3)
    # coefficient scan: scanPosLast of every 8x8 TU of a frame; costCoeffNxN of every 4x4 group
    ntu = (W // 8) * (H // 8)
    co = rng.integers(-40, 41, size=(ntu, 64)).astype(np.int16)
    co[rng.random(co.shape) < 0.85] = 0
    co[:, 0] |= 1
    dco = DevBuf(co)
    sg, fl, nm_, la = DevBuf.zeros((ntu, 64), np.uint16), DevBuf.zeros((ntu, 64), np.uint16), DevBuf.zeros((ntu, 64), np.uint8), DevBuf.zeros((ntu,), np.int32)
    ms = timeit(lambda: check(L.x265hip_scan_pos_last_batch(3, 0, dco.ptr, ntu, sg.ptr, fl.ptr, nm_.ptr, la.ptr, None)))
    row("scanPosLast, 8x8 TUs of a frame", ntu, ms, ntu * (128 + 5 * 4 + 4))
    sb = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "entropy_state_bits.json")))["entropyStateBits"], np.uint32)
    check(L.x265hip_set_entropy_state_bits(sb.ctypes.data_as(C.c_void_p)))
    ncg = ntu * 4
    cj = (CoeffGroupJob * ncg)()
    for i in range(ncg):
        cj[i].coeffOffset, cj[i].trSize, cj[i].scanType, cj[i].scanFlagMask, cj[i].offset, cj[i].scanPosSigOff, cj[i].subPosBase = (i // 4) * 64 + (i % 4 // 2) * 32 + (i % 2) * 4, 8, 0, 0x5a5a, 9, 15, (i % 4) * 16
    dcj = DevBuf(np.frombuffer(bytes(cj), np.uint8).copy())
    ctx, absC, bits = DevBuf(rng.integers(2, 125, size=(ncg, 64)).astype(np.uint8)), DevBuf.zeros((ncg, 16), np.uint16), DevBuf.zeros((ncg,), np.uint32)
    ms = timeit(lambda: check(L.x265hip_cost_coeff_nxn_batch(dco.ptr, dcj.ptr, ncg, ctx.ptr, 64, absC.ptr, bits.ptr, None)))
    row("costCoeffNxN, 4x4 groups of a frame", ncg, ms, ncg * (32 + 32 + 16 + 4))
    # weighted bi-prediction of every 16x16 PU (luma + chroma)
    HC, SC = (H + 2 * M) // 2, S // 2
    planes = [DevBuf(pic), DevBuf(pic[::2, ::2].copy()), DevBuf(pic[1::2, ::2].copy())]
    dst = [DevBuf.zeros(pic.shape, np.uint8), DevBuf.zeros((HC, SC), np.uint8), DevBuf.zeros((HC, SC), np.uint8)]
    ya = YuvStruct(planes[0].ptr, planes[1].ptr, planes[2].ptr, S, SC)
    yd = YuvStruct(dst[0].ptr, dst[1].ptr, dst[2].ptr, S, SC)
    pus = [(M + x, M + y) for y in range(0, H - 15, 16) for x in range(0, W, 16)]
    dxy = dev_i32(np.array(pus, np.int32).reshape(-1))
    mv = dev_i32(rng.integers(-30, 31, size=2 * len(pus)).astype(np.int32))
    wp = (WeightParam * 3)(WeightParam(70, 5, 6, 1), WeightParam(60, -3, 6, 1), WeightParam(64, 0, 6, 0))
    ms = timeit(lambda: check(L.x265hip_motion_compensation_batch(depth, 16, 16, C.byref(ya), C.byref(ya), C.byref(yd), dxy.ptr, mv.ptr, mv.ptr, len(pus), wp, wp, None)))
    row("motionCompensation weighted bi, 16x16 PUs", len(pus), ms, len(pus) * (2 * (23 * 23 + 2 * 11 * 11) + 384))
    # lookahead: AQ block energies of the frame, 16 candidate weight costs on the lowres planes, one CU-tree step
    en, sums = DevBuf.zeros(((W // 16 + 1) * (H // 16 + 1),), np.uint32), DevBuf.zeros((6,), np.uint64)
    yo = YuvStruct(planes[0].ptr + org, planes[1].ptr + (M // 2) * SC + M // 2, planes[2].ptr + (M // 2) * SC + M // 2, S, SC)
    ms = timeit(lambda: check(L.x265hip_aq_block_energy(depth, C.byref(yo), W, H, 16, en.ptr, sums.ptr, None)))
    row("AQ block energy, 16x16 groups of a frame", (W // 16) * ((H + 15) // 16), ms, W * H * 3 // 2)
    lw, lh, lm = 960, 544, 64
    ls = lw + 2 * lm
    lo = DevBuf(rng.integers(0, 256, size=(lh + 2 * lm, ls)).astype(np.uint8))
    ic = DevBuf(rng.integers(100, 3000, size=(lw // 8) * (lh // 8)).astype(np.int32))
    cands = (WeightParam * 16)(*[WeightParam(int(rng.integers(30, 127)), int(rng.integers(-20, 21)), 6, 1) for _ in range(16)])
    costs = DevBuf.zeros((16,), np.uint32)
    ms = timeit(lambda: check(L.x265hip_lookahead_weight_cost_batch(depth, lo.at(lm * ls + lm), lo.at(lm * ls + lm + 3), ls, lw, lh, ic.ptr, cands, 16, costs.ptr, None)))
    row("weightCostLuma, 16 candidate weights (lowres 960x544)", 16, ms, 16 * 2 * lw * lh)
    wcu, hcu = lw // 8, lh // 8
    ncu = wcu * hcu
    pin, lc = DevBuf(rng.integers(0, 30000, size=ncu).astype(np.uint16)), DevBuf((rng.integers(0, 4000, size=ncu) | (rng.integers(1, 4, size=ncu) << 14)).astype(np.uint16))
    iq, m0 = DevBuf(rng.integers(100, 900, size=ncu).astype(np.int32)), dev_i32(rng.integers(-60, 61, size=2 * ncu).astype(np.int32))
    r0, r1, scr = DevBuf.zeros((ncu,), np.uint16), DevBuf.zeros((ncu,), np.uint16), DevBuf.zeros((2 * ncu,), np.uint64)
    ms = timeit(lambda: check(L.x265hip_cutree_propagate(wcu, hcu, 30, 1, 0.033, 1, 2, 1, 0, pin.ptr, ic.ptr, lc.ptr, iq.ptr, m0.ptr, m0.ptr, r0.ptr, r1.ptr, scr.ptr, None)))
    row("CU-tree propagate step (1080p block grid)", ncu, ms, ncu * (2 + 4 + 2 + 4 + 16 + 16))
    for r in res:
        print("%-58s n %8d  %9.2f us  %8.1f GB/s" % (r["kernel"], r["n"], r["us"], r["GBps"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
