#!/usr/bin/env python3
"""Interleaved A/B runs of the bound encoder (integration/_build/x265_hip_8bit) on the bench clip: every configuration (a set of X265HIP_* environment
variables) is run once per round, rounds alternate, so that drift of the box hits all configurations alike.  Prints per configuration the mean /
median / best fps (the CLI's own fps line), user CPU seconds, and whether every bitstream was byte-identical to the unmodified reference's.
    python tools/ab_encode.py --rounds 5 --frames 120 base: sad0:X265HIP_SADPLANES=0 l12:X265HIP_SADPLANES_LEVELS=12
"""
import argparse
import hashlib
import json
import os
import resource
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
INTEG = os.path.join(ROOT, "integration", "_build")


def run(exe, args, out, env):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    p = subprocess.run([exe] + args + ["-o", out], capture_output=True, text=True, env=env, timeout=1200)
    wall = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    fps = None
    for line in (p.stderr + p.stdout).splitlines():
        if line.startswith("encoded") and "fps" in line:
            fps = float(line.split("(")[1].split("fps")[0])
    return {"rc": p.returncode, "fps": fps, "wall": wall, "user": r1.ru_utime - r0.ru_utime, "sys": r1.ru_stime - r0.ru_stime,
            "served": [l for l in p.stderr.splitlines() if l.startswith("x265hip:")], "sha": hashlib.sha256(open(out, "rb").read()).hexdigest()[:16] if p.returncode == 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", help="name:VAR=val,VAR=val (name: alone = defaults)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--preset", default="medium")
    ap.add_argument("--extra", default="--me hex")
    ap.add_argument("--out", default=None)
    ap.add_argument("--bits", type=int, default=8, help="encoder build: 8, 10 or 12 (x265_<bits>bit / x265_hip_<bits>bit); the clip stays 8-bit input")
    a = ap.parse_args()
    w, h = map(int, a.res.split("x"))
    from x265_amd.synth import make_clip
    clip = "/tmp/ab_clip_%dx%d_%d.yuv" % (w, h, a.frames)
    if not os.path.exists(clip):
        make_clip(clip, w, h, a.frames, seed=4321)
    args = ["--input", clip, "--input-res", a.res, "--input-depth", "8", "--fps", "30", "--frames", str(a.frames), "--preset", a.preset, "--hash", "1"] + a.extra.split()
    ref = run(os.path.join(REF, "x265_%dbit" % a.bits), args, "/tmp/ab_ref.hevc", dict(os.environ))
    cfgs = []
    for c in a.configs:
        name, _, rest = c.partition(":")
        env = dict(kv.split("=", 1) for kv in rest.split(";" if ";" in rest or rest.count("=") == 1 else ",") if kv)
        cfgs.append((name, env))
    res = {name: [] for name, _ in cfgs}
    for r in range(a.rounds):
        for name, env in (cfgs if r % 2 == 0 else cfgs[::-1]):
            e = dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1", **env)
            # a configuration named REF* is the unmodified reference encoder, VEC* the reference with its intrinsics transforms (--asm SSE4.1), interleaved like the rest
            if name.startswith("REF"):
                res[name].append(run(os.path.join(REF, "x265_%dbit" % a.bits), args, "/tmp/ab_%s.hevc" % name, e))
            elif name.startswith("VEC"):
                res[name].append(run(os.path.join(REF, "x265_vec_8bit"), args + ["--asm", "SSE4.1"], "/tmp/ab_%s.hevc" % name, e))
            else:
                res[name].append(run(os.path.join(INTEG, "x265_hip_%dbit" % a.bits), args, "/tmp/ab_%s.hevc" % name, e))
    summary = {"clip": "%s %d frames preset %s %s" % (a.res, a.frames, a.preset, a.extra), "reference": {"fps": ref["fps"], "user": round(ref["user"], 1)}, "configs": {}}
    print("reference: %.2f fps, user %.1f s" % (ref["fps"], ref["user"]))
    for name, env in cfgs:
        f = [x["fps"] for x in res[name] if x["fps"]]
        u = [x["user"] for x in res[name]]
        ident = all(x["sha"] == ref["sha"] for x in res[name])
        summary["configs"][name] = {"env": env, "fps": [round(x, 2) for x in f], "fps_mean": round(statistics.mean(f), 2), "fps_median": round(statistics.median(f), 2),
                                    "fps_best": round(max(f), 2), "user_mean": round(statistics.mean(u), 1), "byte_identical": ident, "served": res[name][-1]["served"]}
        print("%-14s fps mean %.2f median %.2f best %.2f  user %.1f s  identical %s   %s" % (name, statistics.mean(f), statistics.median(f), max(f), statistics.mean(u), ident,
                                                                                              " ".join("%.1f" % x for x in f)))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(summary, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
