#!/usr/bin/env python3
"""The search-window kernel (x265_amd/csrc/sadsurf.hip, sadsurf_ctu_kernel) on its own, through the C ABI it is used through: a 1920x1080 reference
picture mirror (x265hip_refpic_*), K source pictures (x265hip_srcpic_*), one SAD surface per source (x265hip_sadsurf_attach).  The library times
every launch with HIP events on the stream it runs on (x265hip_sadsurf_stats); this script only arranges WHEN rows become available:

  frame    the whole reference picture is final, then the K surfaces are attached one after the other: K launches of all 17 CTU rows (510 CTUs)
  batch    the K surfaces are attached first, then the picture becomes final at once: one launch of K x 510 CTUs (rows of several surfaces share it)
  bands    the K surfaces are attached first, then the picture arrives in bands of 64 rows as the encoder reconstructs it: 17 launches of K x 30 CTUs

Prints one JSON line per mode: launches, CTUs per launch, microseconds per launch, absolute differences per second against the measured issue
ceiling of v_qsad_pk_u16_u8 (the roofline that binds this kernel: tools/micro/qsad_rate), and the SURVEY.md §8d algorithmic bytes per second.
§8d's "batched exhaustive search of one block over an R x R window counts the unique footprint" is applied to what a workgroup stages — ONE
64x64 source block and ONE (64 + R - 1)^2 window per CTU (R = 2 S = 64, B = 1), shared by the CTU's 16 + 4 + 1 block searches — plus the bytes
the CTU really emits (a 16 x 16 window of entries and an origin per block; the surfaces stay in LDS): 4096 + 16 129 + 13 396 = 33 621 bytes
per complete CTU.  (Rounds 1-3 charged the footprint 21 times per CTU with a 4 R^2 result term that is never written: 508 437 bytes, 15 times
too many — VERDICT r03.)  Used by bench.py for the `roofline` block and by tools/collect_profiles.sh under rocprofv3."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MX, MY = 96, 80            # PicYuv's luma margins at CTU 64 (picyuv.cpp:87-89)
HBM_PEAK = 8000.0


VALU_SAD_CEILING_T = 95.2     # T absolute differences per second: v_qsad_pk_u16_u8 issue rate of the whole chip, tools/micro/qsad_rate (profiles/r03_v2_sadsurf_kernel.txt)
EMITTED_PER_CTU = 16 * (512 + 4) + 4 * (1024 + 4) + (1024 + 4)     # 16x16 level: u16 entries; 32x32 and 64x64: u32; 4 bytes of origin per block


def unit_bytes(S):
    """SURVEY 8d unique footprint of the staged unit (one CTU) + what it emits"""
    R = 2 * S
    return 64 * 64 + (64 + R - 1) ** 2 + EMITTED_PER_CTU


def pictures(w, h, k, seed):
    from x265_amd.synth import make_scene
    sc = make_scene(w, h, 8, seed=seed)
    ref = sc["ref"]
    rows = ((h + 63) // 64) * 64 + 2 * MY
    stride = ((w + 2 * MX + 63) // 64) * 64
    buf = np.zeros((rows, stride), np.uint8)
    buf[:h + 2 * MY, :w + 2 * MX] = np.pad(ref, ((MY, MY), (MX, MX)), mode="edge")
    srcs = [np.ascontiguousarray(np.roll(sc["src"], (3 * i, -5 * i), axis=(0, 1))) for i in range(k)]
    return buf, stride, rows, srcs


def run_mode(hp, L, mode, w, h, k, S, reps, buf, stride, rows, srcs):
    def stats():
        v = [C.c_uint64() for _ in range(4)]
        L.x265hip_sadsurf_stats(*[C.byref(x) for x in v])
        return [x.value for x in v]
    sps = []
    for s in srcs:
        sp = L.x265hip_srcpic_create(8, w, h)
        assert sp, L.x265hip_last_error()
        hp.check(L.x265hip_srcpic_upload(sp, s.ctypes.data, s.shape[1]))
        sps.append(sp)
    before = None
    for rep in range(reps + 1):
        if rep == 1:
            before = stats()                      # the first repetition is the warm-up
        rp = L.x265hip_refpic_create(8, w, h, stride, MX, MY, rows, buf.ctypes.data)
        assert rp, L.x265hip_last_error()
        sss = []
        if mode == "frame":
            hp.check(L.x265hip_refpic_rows_final(rp, h))
            hp.check(L.x265hip_refpic_wait(rp))
        for sp in sps:
            ss = L.x265hip_sadsurf_attach(sp, rp, S, 180)
            assert ss, L.x265hip_last_error()
            sss.append(ss)
            if mode == "frame":
                hp.check(L.x265hip_refpic_wait(rp))
        if mode == "batch":
            hp.check(L.x265hip_refpic_rows_final(rp, h))
        elif mode == "bands":
            for r in list(range(64, h, 64)) + [h]:
                hp.check(L.x265hip_refpic_rows_final(rp, r))
                hp.check(L.x265hip_refpic_wait(rp))
        hp.check(L.x265hip_refpic_wait(rp))
        for ss in sss:
            v = C.cast(L.x265hip_sadsurf_get_view(ss), C.POINTER(hp.SadSurfView)).contents
            assert v.ctuRowsReady[0] == (h + 63) // 64
            L.x265hip_sadsurf_release(ss)
        L.x265hip_refpic_wait(rp)
        L.x265hip_refpic_destroy(rp)
    after = stats()
    for sp in sps:
        L.x265hip_srcpic_destroy(sp)
    rows_built = after[1] - before[1]
    launches = after[2] - before[2]
    ns = after[3] - before[3]
    ctus = rows_built * ((w + 63) // 64)
    return {"mode": mode, "surfaces": k, "launches": launches, "ctus_per_launch": round(ctus / launches, 1), "us_per_launch": round(ns / launches / 1e3, 2),
            "us_per_ctu_row": round(ns / rows_built / 1e3, 2), "ctus": ctus, "kernel_ns": ns}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="frame,batch,bands")
    ap.add_argument("--surfaces", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--range", type=int, default=32)
    ap.add_argument("--res", default="1920x1080")
    a = ap.parse_args()
    import x265_amd.hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    w, h = map(int, a.res.split("x"))
    buf, stride, rows, srcs = pictures(w, h, a.surfaces, 4321)
    ub = unit_bytes(a.range)
    for mode in a.modes.split(","):
        r = run_mode(hp, L, mode, w, h, a.surfaces, a.range, a.reps, buf, stride, rows, srcs)
        secs = r["kernel_ns"] * 1e-9
        r["algorithmic_bytes_per_ctu"] = ub
        r["achieved_GBps"] = round(r["ctus"] * ub / secs / 1e9, 1)
        r["frac_of_hbm_peak"] = round(r["ctus"] * ub / secs / 1e9 / HBM_PEAK, 4)
        # 16 blocks x (2 S)^2 vectors x 256 absolute differences per CTU
        r["abs_diff_per_s_T"] = round(r["ctus"] * 16 * (2 * a.range) ** 2 * 256 / secs / 1e12, 2)
        r["frac_of_valu_sad_ceiling"] = round(r["abs_diff_per_s_T"] / VALU_SAD_CEILING_T, 4)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
