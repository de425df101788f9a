"""How long does the HOST need to enqueue one frame pass (13 launches through the C ABI)?  Guides the hipGraph decision."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import hipprim as hp                      # noqa: E402
from x265_amd.framepass import FramePass, Picture       # noqa: E402
from x265_amd.synth import make_scene_yuv               # noqa: E402

L = hp.lib()
hp.check(L.x265hip_init(0))
W, H = 1920, 1080
sc = make_scene_yuv(W, H)
src = Picture(W, H, 8, sc["src"], sc["src_cb"], sc["src_cr"])
ref = Picture(W, H, 8, sc["ref"], sc["ref_cb"], sc["ref_cr"])
pred, rec = Picture(W, H, 8), Picture(W, H, 8)
fp = FramePass(W, H)
for _ in range(5):
    fp.run_yuv(src, ref, pred, rec)
hp.check(L.x265hip_stream_sync(None))
n = 200
t0 = time.perf_counter()
for _ in range(n):
    fp.run_yuv(src, ref, pred, rec)
t1 = time.perf_counter()
hp.check(L.x265hip_stream_sync(None))
t2 = time.perf_counter()
print("host enqueue per frame pass: %.1f us; GPU-bound total per pass: %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
