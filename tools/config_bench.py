"""The other BASELINE.json configurations on one GPU (not the contract bench: bench.py stays on configs[1]).

    python tools/config_bench.py --encode configs2 [--rounds 5] [--frames 24]     # the REAL encode of configs[2] (3840x2160 preset slow --me star --merange 57) or
    python tools/config_bench.py --encode configs3 [--rounds 5] [--frames 8]      # configs[3] (3840x2160 Main10 preset slower --rd 6): bound encoder and reference,
                                                                                   # interleaved rounds, ONE JSON line shaped like bench.py's (value, cpu_baseline,
                                                                                   # byte identity, what the GPU served)
    python tools/config_bench.py --width 3840 --height 2160 --depth 8 --me 3 --subme 3     # the frame-pass harness (rounds 1-2) at configs[2]'s geometry: F chained
                                                                                            # frame passes in flight on F streams, HIP-event timing over `steps` steps
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


ENCODES = {
    "configs2": dict(res="3840x2160", bits=8, preset="slow", extra=["--me", "star", "--merange", "57"], frames=24, input_depth=8,
                     name="BASELINE configs[2]: 3840x2160 preset slow --me star --merange 57"),
    "configs3": dict(res="3840x2160", bits=10, preset="slower", extra=["--rd", "6"], frames=8, input_depth=10,
                     name="BASELINE configs[3]: 3840x2160 Main10 preset slower --rd 6"),
    # (configs[4] names eight GPUs; this is its picture size and preset on ONE: the single-GPU point of that shape — bench.py --gpus N carries the N-GPU form)
    "configs4": dict(res="7680x4320", bits=8, preset="medium", extra=["--me", "hex"], frames=8, input_depth=8,
                     name="BASELINE configs[4]'s shape on one GPU: 7680x4320 preset medium --me hex"),
}


def encode_line(which, rounds, frames, threads):
    """the real encode of one BASELINE configuration: `rounds` interleaved runs of the bound encoder and of the unmodified reference (same arguments)"""
    import statistics
    import ab_encode as ab
    from x265_amd.synth import make_clip
    c = ENCODES[which]
    frames = frames or c["frames"]
    w, h = map(int, c["res"].split("x"))
    clip = "/tmp/cfg_%s_%d.yuv" % (which, frames)
    if not os.path.exists(clip):
        make_clip(clip, w, h, frames, seed=4321, depth=c["input_depth"])
    args = ["--input", clip, "--input-res", c["res"], "--input-depth", str(c["input_depth"]), "--fps", "30", "--frames", str(frames), "--preset", c["preset"], "--hash", "1"] + c["extra"] + threads
    hip, ref = os.path.join(ab.INTEG, "x265_hip_%dbit" % c["bits"]), os.path.join(ab.REF, "x265_%dbit" % c["bits"])
    runs = {"hip": [], "ref": []}
    for r in range(rounds):
        for k in (("hip", "ref") if r % 2 == 0 else ("ref", "hip")):
            env = dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1") if k == "hip" else dict(os.environ)
            runs[k].append(ab.run(hip if k == "hip" else ref, args, "/tmp/cfg_%s_%s.hevc" % (which, k), env))
    ok = all(x["rc"] == 0 for k in runs for x in runs[k])
    fps = {k: [frames / x["wall"] for x in runs[k]] for k in runs}
    flags = None
    try:
        flags = dict(l.strip().split("=", 1) for l in open(os.path.join(ab.REF, "build_flags.txt")) if "=" in l).get("REF_OPT")
    except OSError:
        pass
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        quota = None
    out = {"metric": "encode fps (%s)" % c["name"], "value": round(statistics.mean(fps["hip"]), 3) if ok else None, "unit": "frames/s", "n_gpus": 1, "rounds": rounds,
           "frames": frames, "higher_is_better": True, "dtype": "u8" if c["bits"] == 8 else "u16", "data": "synthetic",
           "config": {"workload": c["name"] + ", synthetic clip of x265_amd/synth.make_clip (seed 4321), %d frames; all frames / wall clock of the encoder process, mean of %d "
                                              "interleaved rounds" % (frames, rounds), "args": " ".join(args[2:]), "fps_per_round": [round(x, 3) for x in fps["hip"]],
                      "cli_fps": [x["fps"] for x in runs["hip"]], "user_cpu_s": round(statistics.mean(x["user"] for x in runs["hip"]), 1),
                      "served_by_gpu": runs["hip"][-1]["served"]},
           "cpu_baseline": {"value": round(statistics.mean(fps["ref"]), 3) if ok else None, "unit": "frames/s", "kind": "reference", "build_flags": flags,
                            "cores": int(min(os.cpu_count() or 1, quota)) if quota else os.cpu_count(), "fps_per_round": [round(x, 3) for x in fps["ref"]],
                            "cli_fps": [x["fps"] for x in runs["ref"]], "user_cpu_s": round(statistics.mean(x["user"] for x in runs["ref"]), 1),
                            "sample": "the same clip and arguments through oracle/_ref/x265_%dbit (the unmodified reference, [noasm] C primitives)" % c["bits"],
                            "byte_identical_to_gpu_path": ok and all(x["sha"] == runs["ref"][0]["sha"] for k in runs for x in runs[k])}}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encode", choices=sorted(ENCODES), help="the real encode of a BASELINE configuration (bound encoder + reference) instead of the frame-pass harness")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--threads", default="", help='extra thread arguments for both encoders, e.g. "--pools 24 --frame-threads 6"')
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
    if a.encode:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        encode_line(a.encode, a.rounds, a.frames, a.threads.split())
        return
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
