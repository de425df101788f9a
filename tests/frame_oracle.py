"""CPU checker for the frame pass: ctypes front end of oracle/x265_oracle_frame.c + the seeded scene generator.
TEST INFRASTRUCTURE ONLY (used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from x265_amd.synth import make_scene  # noqa: E402,F401  (seeded input generator, shared with bench.py)

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
    fn.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_ssize_t] * 4 + [C.c_int, C.c_int] + [pp] * 6
    return fn


def oracle_frame_pass(src, ref, depth=8, qp=28, merange=57, method=1, subme=2):
    """Run the C restatement of the frame pass; returns the same dict x265_amd.framepass.FramePass.run_host returns."""
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
    _proto(L, depth)(w, h, depth, qp, merange, method, subme, org(psrc), S, org(pref), S, org(pred), S, org(recon), S, m, m,
                     arr(mv), arr(cost), arr(sa8d), arr(level), arr(numsig), arr(dist))
    return {"mv": mv, "cost": cost, "sa8d": sa8d, "level": level, "numSig": numsig, "dist": dist,
            "pred": np.ascontiguousarray(pred[m:m + h, m:m + w]), "recon": recon}


def same_results(got, want):
    """List of the output names that differ (empty = bit-exact)."""
    bad = []
    for k in ("mv", "cost", "sa8d", "level", "numSig", "dist"):
        for i, (a, b) in enumerate(zip(got[k], want[k])):
            if not np.array_equal(a, b):
                bad.append("%s[%d] (%d of %d rows differ)" % (k, i, int(np.any(a.reshape(len(a), -1) != b.reshape(len(b), -1), axis=1).sum()), len(a)))
    for k in ("pred", "recon"):
        if not np.array_equal(got[k], want[k]):
            bad.append("%s (%d pixels)" % (k, int((got[k] != want[k]).sum())))
    return bad
