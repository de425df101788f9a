#!/usr/bin/env python3
"""The MFMA transform kernels' ceiling on file: x265hip_residual_chain_batch, x265hip_dct_batch and x265hip_idct_batch at
batch sizes from one 1080p frame's TUs up to >= 1 M coefficients x 32, HIP events on the launch stream, algorithmic bytes per SURVEY.md §8d
(chain: N^2 (3B + 2); dct / idct: 4 N^2).  Prints one JSON line; run it under rocprofv3 --kernel-trace --stats for the per-kernel averages.
    python tools/chain_bench.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check
    L = hp.lib()
    check(L.x265hip_init(0))
    depth, N = 8, 32
    st = C.c_void_p()
    check(L.x265hip_stream_create(C.byref(st)))
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    check(L.x265hip_event_create(C.byref(ev0)))
    check(L.x265hip_event_create(C.byref(ev1)))

    def timed(fn, iters=20, warm=3):
        for _ in range(warm):
            fn()
        check(L.x265hip_event_record(ev0, st))
        for _ in range(iters):
            fn()
        check(L.x265hip_event_record(ev1, st))
        ms = C.c_float()
        check(L.x265hip_event_elapsed_ms(ev0, ev1, C.byref(ms)))
        return ms.value / iters

    out = {"depth": depth, "tu": N, "unit": "GB/s algorithmic", "rows": []}
    rng = np.random.default_rng(3)
    for n in (1980, 8192, 32768, 131072):                      # one 1080p frame's 32x32 TUs ... 134 M coefficients
        side = int(np.ceil(np.sqrt(n)))
        W = side * N
        H = ((n + side - 1) // side) * N
        fenc = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        pred = np.clip(fenc.astype(np.int32) + rng.integers(-12, 13, size=(H, W)), 0, 255).astype(np.uint8)
        df, dp, dr = DevBuf(fenc), DevBuf(pred), DevBuf.zeros((H, W), np.uint8)
        off = np.array([(i // side) * N * W + (i % side) * N for i in range(n)], np.int32)
        do = DevBuf(off)
        qc = DevBuf(np.full(N * N, 16 << 4, np.int32))
        level, ns, dist = DevBuf.zeros((n, N * N), np.int16), DevBuf.zeros((n,), np.uint32), DevBuf.zeros((n,), np.uint64)
        qp = 28
        ts = 15 - depth - 5
        qbits = 14 + qp // 6 + ts
        args = (N, depth, df.ptr, W, dp.ptr, W, dr.ptr, W, do.ptr, do.ptr, do.ptr, qc.ptr, qbits, 85 << (qbits - 9), [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6),
                20 - 14 - ts, level.ptr, ns.ptr, dist.ptr, n, st)
        t_chain = timed(lambda: check(L.x265hip_residual_chain_batch(*args)))
        resi = DevBuf(rng.integers(-255, 256, size=(n, N * N)).astype(np.int16))
        coef = DevBuf.zeros((n, N * N), np.int16)
        lin = DevBuf(np.arange(n, dtype=np.int32) * N * N)
        t_dct = timed(lambda: check(L.x265hip_dct_batch(N, 0, depth, resi.ptr, N, lin.ptr, coef.ptr, n, st)))
        t_idct = timed(lambda: check(L.x265hip_idct_batch(N, 0, depth, coef.ptr, resi.ptr, N, lin.ptr, n, st)))
        px = n * N * N
        out["rows"].append({"tus": n, "coefficients": px, "chain_ms": round(t_chain, 4), "chain_GBps": round(px * 5 / t_chain / 1e6, 1),
                            "dct_ms": round(t_dct, 4), "dct_GBps": round(px * 4 / t_dct / 1e6, 1), "idct_ms": round(t_idct, 4), "idct_GBps": round(px * 4 / t_idct / 1e6, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
