#!/usr/bin/env python3
"""BASELINE.json's metric as the reference reports it: the CLI's own "encoded N frames in Xs (Y fps)" line, for
  * oracle/_ref/x265_8bit      the unmodified reference encoder ([noasm] C primitives; no nasm in the image), and
  * integration/_build/x265_hip_8bit  the same objects + x265_amd/host/*.cpp + libx265hip.so (the lookahead seam on the GPU),
on the same synthetic 1080p clip (x265_amd/synth.make_clip, seed 4321), same arguments, all host cores; the two bitstreams must be
byte-identical.  Used by bench.py (`encode_fps` / `reference_encoder` keys) and runnable on its own:
    python tools/encode_fps.py --frames 60 [--preset medium] [--res 1920x1080] [--extra "--me hex"]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")                 # the reference alone (CPU baseline) and the emulated-ABI test binaries
INTEG = os.path.join(ROOT, "integration", "_build")     # the product: reference objects + x265_amd/host/*.cpp + libx265hip.so


def _run(exe, args, out, env=None, timeout=1200):
    t0 = time.perf_counter()
    p = subprocess.run([exe] + args + ["-o", out], capture_output=True, text=True, timeout=timeout, env=env)
    wall = time.perf_counter() - t0
    fps, log = None, p.stderr + p.stdout
    for line in log.splitlines():
        if line.startswith("encoded") and "fps" in line:
            fps = float(line.split("(")[1].split("fps")[0])
    served = [l for l in log.splitlines() if l.startswith("x265hip:")]
    return {"rc": p.returncode, "fps": fps, "wall_s": round(wall, 2), "served": served, "tail": log[-400:] if p.returncode else ""}


def measure(frames=60, width=1920, height=1080, bits=8, preset="medium", extra=("--me", "hex"), seed=4321, keep=False, clip=None, repeat=1, input_depth=8):
    """input_depth 10: the clip holds 16-bit little-endian samples with 10 significant bits (PicYuv::copyFromPicture takes its 16-bit path; Main10 / Main12 builds)."""
    ref, hip = os.path.join(REF, "x265_%dbit" % bits), os.path.join(INTEG, "x265_hip_%dbit" % bits)
    if not (os.path.exists(ref) and os.path.exists(hip)):
        return {"error": "oracle/_ref/x265_%dbit / integration/_build/x265_hip_%dbit not built (make -C oracle ref; make -C integration hip where /root/reference exists)" % (bits, bits)}
    from x265_amd.synth import make_clip
    path = clip or "/tmp/x265hip_clip_%dx%d_%d_%d_d%d.yuv" % (width, height, frames, seed, input_depth)
    if not os.path.exists(path):
        make_clip(path, width, height, frames, seed=seed, depth=input_depth)
    args = ["--input", path, "--input-res", "%dx%d" % (width, height), "--input-depth", str(input_depth), "--fps", "30", "--frames", str(frames), "--preset", preset,
            "--hash", "1"] + list(extra)
    o_ref, o_hip = "/tmp/x265hip_ref_%d.hevc" % os.getpid(), "/tmp/x265hip_gpu_%d.hevc" % os.getpid()
    res = {"clip": "%dx%d %d-bit-encode, %d synthetic frames (make_clip seed %d)" % (width, height, bits, frames, seed), "cmd": " ".join(args[2:]),
           "host_cores": os.cpu_count()}
    try:
        runs_ref = [_run(ref, args, o_ref) for _ in range(repeat)]
        runs_hip = [_run(hip, args, o_hip, env=dict(os.environ, X265HIP_VERBOSE="1", X265HIP="require")) for _ in range(repeat)]
        best = lambda rs: max(rs, key=lambda r: r["fps"] or 0)  # noqa: E731
        r_ref, r_hip = best(runs_ref), best(runs_hip)
        res["reference"] = {k: r_ref[k] for k in ("fps", "wall_s", "rc")}
        res["gpu"] = {k: r_hip[k] for k in ("fps", "wall_s", "rc", "served")}
        if r_ref["rc"] or r_hip["rc"]:
            res["error"] = (r_ref["tail"] or r_hip["tail"])[-300:]
            return res
        h = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in (o_ref, o_hip)]
        res["bitstream_bytes"] = os.path.getsize(o_ref)
        res["byte_identical"] = h[0] == h[1]
        res["sha256"] = h[0][:16]
    finally:
        for p in (o_ref, o_hip) + (() if keep or clip else (path,)):
            if os.path.exists(p):
                os.remove(p)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--preset", default="medium")
    ap.add_argument("--extra", default="--me hex")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    w, h = (int(v) for v in a.res.split("x"))
    print(json.dumps(measure(a.frames, w, h, a.bits, a.preset, tuple(a.extra.split()), keep=a.keep, repeat=a.repeat)))
