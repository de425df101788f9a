"""Lookahead front end on one GPU at 1080p: Lowres::init (downscale + 4 extended hpel planes) and the intra estimate of every
8x8 lowres block.  HIP-event timing, algorithmic GB/s; the CPU oracle of the same pass beside it.  Not the contract bench."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from x265_amd import hipprim as hp                      # noqa: E402
from x265_amd.hipprim import DevBuf, check              # noqa: E402
from x265_amd.synth import make_scene                   # noqa: E402


def timeit(fn, iters=50, warm=5):
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
    depth = int(os.environ.get("DEPTH", "8"))
    B = 1 if depth == 8 else 2
    W, H, M = 1920, 1080, 96
    pic = make_scene(W, H, depth, seed=4321)["src"]
    src = np.ascontiguousarray(np.pad(pic, ((M, M), (M, M + 8)), mode="edge"))     # PicYuv margins, borders extended
    S = src.shape[1]
    lw, lh = ((W // 2 + 7) // 8) * 8, ((H // 2 + 7) // 8) * 8
    ls = lw + 2 * M
    ls += (32 - ls % 32) % 32
    pe = (lh + 2 * M) * ls
    org = M * ls + M
    ds = DevBuf(src)
    planes = DevBuf.zeros((4, lh + 2 * M, ls), src.dtype)
    ptrs = (C.c_void_p * 4)(*[planes.at(i * pe + org) for i in range(4)])
    wcu, hcu = lw // 8, lh // 8
    cost, mode = DevBuf.zeros((wcu * hcu,), np.int32), DevBuf.zeros((wcu * hcu,), np.uint8)
    rows, est = DevBuf.zeros((hcu,), np.int32), DevBuf.zeros((1,), np.int32)
    srcp = ds.at(M * S + M)

    t_init = timeit(lambda: check(L.x265hip_lowres_init(depth, srcp, S, ptrs, ls, lw, lh, M, M, None)))
    t_intra = timeit(lambda: check(L.x265hip_lowres_intra_estimate(depth, planes.at(org), ls, wcu, hcu, cost.ptr, mode.ptr, rows.ptr, est.ptr, None)))
    init_bytes = (2 * lw + 1) * (2 * lh + 1) * B + 4 * pe * B                 # source once + four padded planes written
    intra_bytes = lw * lh * B + wcu * hcu * 5                               # plane once + cost/mode out
    out = {"workload": "1920x1080 %d-bit -> %dx%d lowres, %d blocks" % (depth, lw, lh, wcu * hcu),
           "lowres_init_ms": round(t_init, 4), "lowres_init_GBps": round(init_bytes / t_init / 1e6, 1),
           "intra_estimate_ms": round(t_intra, 4), "intra_estimate_GBps": round(intra_bytes / t_intra / 1e6, 1),
           "frames_per_s": round(1000.0 / (t_init + t_intra), 1), "costEst": int(est.get()[0])}
    # ---- P-frame cost pass: NP (frame, reference) pairs per launch, x265's default slicing at 1080p (6 slices of 10 rows)
    NP = int(os.environ.get("PAIRS", "64"))
    rng = np.random.default_rng(7)
    ref_pic = make_scene(W, H, depth, seed=4321)["ref"]
    rsrc = np.ascontiguousarray(np.pad(ref_pic, ((M, M), (M, M + 8)), mode="edge"))
    drs = DevBuf(rsrc)
    rplanes = DevBuf.zeros((4, lh + 2 * M, ls), src.dtype)
    rptrs = (C.c_void_p * 4)(*[rplanes.at(i * pe + org) for i in range(4)])
    check(L.x265hip_lowres_init(depth, drs.at(M * S + M), S, rptrs, ls, lw, lh, M, M, None))
    ncu = wcu * hcu
    descs = (hp.LookaheadPair * NP)()
    keep = []
    for i in range(NP):
        o = [DevBuf.zeros((ncu, 2), np.int32), DevBuf.zeros((ncu,), np.int32), DevBuf.zeros((ncu,), np.uint16), DevBuf.zeros((hcu,), np.int32),
             DevBuf.zeros((ncu,), np.uint64)]
        d = descs[i]
        d.fenc, d.ref, d.intraCost = planes.at(org), rplanes.at(org), cost.ptr
        d.mvs, d.mvCosts, d.lowresCosts, d.rowSatds, d.sync = [b.ptr for b in o]
        keep.append(o)
    ddesc = DevBuf(np.frombuffer(bytes(descs), np.uint8))
    estp = DevBuf.zeros((NP, 4), np.int64)
    qp = 12 + 6 * (depth - 8)
    half = 2 * 32768
    tab = np.zeros(2 * half + 1, np.uint16)
    check(L.x265hip_mvcost_table(qp, depth, tab.ctypes.data, half))
    dtab = DevBuf(tab)
    rps = max(hcu // 8, 10)
    nsl = hcu // rps
    ep = [0]

    def run_cost():
        ep[0] += 1
        check(L.x265hip_lookahead_cost_p_batch(depth, ddesc.ptr, NP, ls, pe, wcu, hcu, rps, nsl, dtab.at(half), ep[0], estp.ptr, None))
    t_cost = timeit(run_cost, iters=10, warm=2)
    out.update({"cost_pairs": NP, "cost_slices": nsl, "cost_ms_per_launch": round(t_cost, 3), "cost_pairs_per_s": round(NP * 1000.0 / t_cost, 1),
                "cost_blocks_per_s": round(NP * ncu * 1000.0 / t_cost), "cost_est0": int(estp.get()[0, 0])})
    if os.environ.get("CPU", "1") != "0":
        from backends import Orc
        import ctypes
        o = Orc(depth)
        pl0 = o.lowres_pass(rsrc, (M, M), W, H, M, M)[4]
        r1 = o.lowres_pass(src, (M, M), W, H, M, M)
        mvs, mvc = np.zeros((ncu, 2), np.int32), np.zeros(ncu, np.int32)
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        from oracle import pyoracle as po
        refs = (ctypes.c_void_p * 4)(*[po.ptr(p, M, M).value for p in pl0])
        t0 = time.time()
        e = o._f("orc_lookahead_cost_p")(po.ptr(r1[4][0], M, M), refs, ls, wcu, hcu, rps, nsl, depth, po.ptr(r1[1]),
                                         po.vp(tab.ctypes.data + 2 * half), po.ptr(mvs), po.ptr(mvc), po.ptr(lc), po.ptr(rows), po.ptr(imb))
        out["cost_cpu_oracle_ms"] = round((time.time() - t0) * 1000, 1)
        out["cost_cpu_matches"] = bool(int(e) == out["cost_est0"] and np.array_equal(mvs, keep[0][0].get()))
    if os.environ.get("CPU", "1") != "0":
        from backends import Orc
        o = Orc(depth)
        t0 = time.time()
        r = o.lowres_pass(src, (M, M), W, H, M, M)
        out["cpu_oracle_ms"] = round((time.time() - t0) * 1000, 1)
        out["cpu_matches"] = bool(r[0] == out["costEst"] and np.array_equal(r[1], cost.get()) and np.array_equal(r[2], mode.get()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
