"""CPU checker for the frame pass: ctypes front end of oracle/x265_oracle_frame.c + the seeded scene generator.
TEST INFRASTRUCTURE ONLY (used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from x265_amd.synth import make_scene, make_scene_yuv  # noqa: E402,F401  (seeded input generators, shared with bench.py)

MARGIN = 96
CU_SIZES = (64, 32, 16, 8)
TU_SIZES = (32, 8)


def counts(width, height):
    ncu = [(width // s) * (height // s) for s in CU_SIZES]
    n32 = ((width & ~31) // 32) * ((height & ~31) // 32)
    n8 = (width // 8) * (height // 8) - n32 * 16
    return ncu, [n32, n8]


def _proto(L, depth):
    fn = getattr(L, "orc_frame_pass_%s" % po.sfx(depth))
    pp = C.POINTER(C.c_void_p)
    fn.restype = None
    fn.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_ssize_t] * 4 + [C.c_int, C.c_int] + [pp] * 6 + [pp] * 4 + [C.c_ssize_t] * 4 + [pp] * 3 + [C.c_void_p, pp, pp, pp]
    return fn


def oracle_frame_pass(src, ref, depth=8, qp=28, merange=57, method=1, subme=2, src_c=None, ref_c=None, ref1=None, ref1_c=None):
    """Run the C restatement of the frame pass; returns the same dict x265_amd.framepass.FramePass.run_host returns.
    src_c / ref_c: optional (cb, cr) 4:2:0 planes -> the YUV pass (chroma prediction, chroma residual chain).
    ref1 (/ ref1_c): a second reference -> the B pass: list-1 search (outputs mv1 / cost1) and bi-predictive prediction."""
    L = po.oracle()
    h, w = src.shape
    m = MARGIN
    S = w + 2 * m
    psrc = np.ascontiguousarray(np.pad(src, m, mode="edge"))
    pref = np.ascontiguousarray(np.pad(ref, m, mode="edge"))
    pred = np.zeros_like(psrc)
    recon = np.zeros_like(psrc)
    ncu, ntu = counts(w, h)
    mv = [np.zeros((n, 2), np.int32) for n in ncu]
    cost = [np.zeros(n, np.int32) for n in ncu]
    sa8d = [np.zeros(n, np.int32) for n in ncu]
    level = [np.zeros((n, s * s), np.int16) for n, s in zip(ntu, TU_SIZES)]
    numsig = [np.zeros(n, np.uint32) for n in ntu]
    dist = [np.zeros(n, np.uint64) for n in ntu]

    def arr(lst):
        return (C.c_void_p * len(lst))(*[a.ctypes.data for a in lst])
    org = lambda a: C.c_void_p(a.ctypes.data + (m * S + m) * a.itemsize)  # noqa: E731
    yuv = src_c is not None
    mc = m // 2
    Sc = w // 2 + 2 * mc
    corg = lambda a: a.ctypes.data + (mc * Sc + mc) * a.itemsize  # noqa: E731
    if yuv:
        psc = [np.ascontiguousarray(np.pad(p, mc, mode="edge")) for p in src_c]
        prc = [np.ascontiguousarray(np.pad(p, mc, mode="edge")) for p in ref_c]
        ppc = [np.zeros_like(p) for p in psc]
        pcc = [np.zeros_like(p) for p in psc]
        clevel = [np.zeros((ntu[t], (TU_SIZES[t] // 2) ** 2), np.int16) for pl in range(2) for t in range(2)]
        cns = [np.zeros(ntu[t], np.uint32) for pl in range(2) for t in range(2)]
        cdist = [np.zeros(ntu[t], np.uint64) for pl in range(2) for t in range(2)]
        cargs = [(C.c_void_p * 2)(*[corg(a) for a in lst]) for lst in (psc, prc, ppc, pcc)] + [Sc, Sc, Sc, Sc, arr(clevel), arr(cns), arr(cdist)]
    else:
        cargs = [None, None, None, None, 0, 0, 0, 0, None, None, None]
    bargs = [None, None, None, None]
    if ref1 is not None:
        pref1 = np.ascontiguousarray(np.pad(ref1, m, mode="edge"))
        mv1 = [np.zeros((n, 2), np.int32) for n in ncu]
        cost1 = [np.zeros(n, np.int32) for n in ncu]
        pr1c = [np.ascontiguousarray(np.pad(p, mc, mode="edge")) for p in ref1_c] if yuv else None
        bargs = [org(pref1), (C.c_void_p * 2)(*[corg(a) for a in pr1c]) if yuv else None, arr(mv1), arr(cost1)]
    _proto(L, depth)(w, h, depth, qp, merange, method, subme, org(psrc), S, org(pref), S, org(pred), S, org(recon), S, m, m,
                     arr(mv), arr(cost), arr(sa8d), arr(level), arr(numsig), arr(dist), *cargs, *bargs)
    out = {"mv": mv, "cost": cost, "sa8d": sa8d, "level": level, "numSig": numsig, "dist": dist,
           "pred": np.ascontiguousarray(pred[m:m + h, m:m + w]), "recon": recon}
    if ref1 is not None:
        out.update({"mv1": mv1, "cost1": cost1})
    if yuv:
        out.update({"clevel": clevel, "cnumSig": cns, "cdist": cdist,
                    "pred_c": [np.ascontiguousarray(p[mc:mc + h // 2, mc:mc + w // 2]) for p in ppc], "recon_c": pcc})
    return out


def same_results(got, want):
    """List of the output names that differ (empty = bit-exact)."""
    bad = []
    for k in ("mv", "cost", "sa8d", "level", "numSig", "dist") + (("mv1", "cost1") if "mv1" in want else ()):
        for i, (a, b) in enumerate(zip(got[k], want[k])):
            if not np.array_equal(a, b):
                bad.append("%s[%d] (%d of %d rows differ)" % (k, i, int(np.any(a.reshape(len(a), -1) != b.reshape(len(b), -1), axis=1).sum()), len(a)))
    for k in ("pred", "recon"):
        if not np.array_equal(got[k], want[k]):
            bad.append("%s (%d pixels)" % (k, int((got[k] != want[k]).sum())))
    if "clevel" in want:
        for k in ("clevel", "cnumSig", "cdist", "pred_c", "recon_c"):
            for i, (a, b) in enumerate(zip(got[k], want[k])):
                if not np.array_equal(a, b):
                    bad.append("%s[%d]" % (k, i))
    return bad
