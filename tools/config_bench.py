"""Frame-pass throughput at the other BASELINE.json configurations on one GPU (not the contract bench: bench.py stays on
configs[1]).  F chained frame passes in flight on F streams, HIP-event timing over `steps` steps.

    python tools/config_bench.py --width 3840 --height 2160 --depth 8 --me 3 --subme 3     # configs[2]: 4K, star, subme 3
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import hipprim as hp                                   # noqa: E402
from x265_amd.hipprim import check                                   # noqa: E402
from x265_amd.framepass import FramePass, Picture                    # noqa: E402
from x265_amd.synth import make_scene_yuv                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--me", type=int, default=1, help="0 dia, 1 hex, 2 umh, 3 star, 5 full")
    ap.add_argument("--subme", type=int, default=2)
    ap.add_argument("--merange", type=int, default=57)
    ap.add_argument("--qp", type=int, default=28)
    ap.add_argument("--frames-in-flight", type=int, default=3)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--b", action="store_true", help="B pass: second reference, both lists searched, bi-predictive prediction")
    a = ap.parse_args()
    L = hp.lib()
    check(L.x265hip_init(0))
    w, h, d, F = a.width, a.height, a.depth, a.frames_in_flight
    # a pool of differently-moving source pictures, like bench.py: a chain that re-encoded one picture against its own
    # reconstruction would find zero motion everywhere and finish its searches at once
    NPOOL = 3
    pool, sc = [], None
    for i in range(NPOOL):
        sc = make_scene_yuv(w, h, depth=d, seed=4321 + i, sigma=3.0 * (1 << (d - 8)))
        pool.append(Picture(w, h, d, sc["src"], sc["src_cb"], sc["src_cr"]))
    refs = [Picture(w, h, d, sc["ref"], sc["ref_cb"], sc["ref_cr"]) for _ in range(F)]
    preds = [Picture(w, h, d) for _ in range(F)]
    recs = [[Picture(w, h, d), Picture(w, h, d)] for _ in range(F)]
    fps = [FramePass(w, h, depth=d, qp=a.qp, merange=a.merange, method=a.me, subme=a.subme) for _ in range(F)]
    streams = []
    for _ in range(F):
        s = C.c_void_p()
        check(L.x265hip_stream_create(C.byref(s)))
        streams.append(s)
    cur = list(refs)

    def step(k):
        for j in range(F):
            rec = recs[j][k & 1]
            if a.b:
                fps[j].run_yuv_b(pool[(k + j) % NPOOL], cur[j], pool[(k + j + 1) % NPOOL], preds[j], rec, streams[j])    # "future" reference: the next source
            else:
                fps[j].run_yuv(pool[(k + j) % NPOOL], cur[j], preds[j], rec, streams[j])
            cur[j] = rec                                             # each chain references its own previous reconstruction

    def sync():
        for s in streams:
            check(L.x265hip_stream_sync(s))
    for k in range(3):
        step(k)
    sync()
    import time
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "%dx%d %d-bit 4:2:0 me %d subme %d merange %d qp %d%s" % (w, h, d, a.me, a.subme, a.merange, a.qp, " B pass" if a.b else ""),
                      "frames_in_flight": F, "steps": a.steps, "frames_per_s": round(F * a.steps / dt, 1), "ms_per_frame": round(dt * 1e3 / (F * a.steps), 4)}))


if __name__ == "__main__":
    main()
