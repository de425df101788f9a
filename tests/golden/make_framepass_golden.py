"""Golden digests of the frame pass at BASELINE.json's configs[2..4] AS CONFIGURED (search method, range, sub-pel level, bit depth, picture
size), produced by the pinned CPU oracle (oracle/x265_oracle_frame.c) on seeded scenes:

    python tests/golden/make_framepass_golden.py            # writes tests/golden/framepass_configs_golden.json (minutes of CPU, one process per case)

The GPU test (tests/test_framepass.py::test_baseline_configs_match_golden) runs the same scenes through the HIP frame pass and only hashes.
Every output array is digested separately (SHA-256 of its bytes) so a mismatch names the stage."""
import hashlib
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

# name -> (width, height, depth, qp, method, merange, subme, seed, pass)   method: 1 HEX, 3 STAR (x265.h)
CASES = {
    "configs[2] 4K slow star merange 57 subme 3": (3840, 2160, 8, 30, 3, 57, 3, 1002, "yuv"),
    "configs[3] 4K Main10 slower star subme 4": (3840, 2160, 10, 32, 3, 57, 4, 1003, "yuv"),
    "configs[4] 8K medium hex subme 2": (7680, 4320, 8, 28, 1, 57, 2, 1004, "yuv"),
    "configs[1] 1080p medium hex subme 2, B pass": (1920, 1080, 8, 28, 1, 57, 2, 1001, "b"),
    "Main12 1080p hex subme 2": (1920, 1080, 12, 34, 1, 57, 2, 1012, "yuv"),
    "Main12 640x360 star subme 3": (640, 360, 12, 30, 3, 57, 3, 1013, "yuv"),
    "small 328x200 star subme 3 (re-derived on the CPU tier)": (328, 200, 8, 28, 3, 57, 3, 1005, "yuv"),
}


def scene(case):
    from x265_amd.synth import make_scene_yuv
    w, h, depth, qp, method, merange, subme, seed, kind = case
    sig = 3.0 * (1 << (depth - 8))
    sc = make_scene_yuv(w, h, depth=depth, seed=seed, tile=48 if w < 1000 else 96, sigma=sig)
    nxt = None
    if kind == "b":
        n = make_scene_yuv(w, h, depth=depth, seed=seed, tile=48 if w < 1000 else 96, sigma=sig * 5 / 3, vmax=5)
        nxt = (n["src"], n["src_cb"], n["src_cr"])
    return sc, nxt


def digest(res):
    """{output name: sha256} of a frame-pass result dict (x265_amd.framepass.FramePass.run_host_yuv[_b] / frame_oracle.oracle_frame_pass)."""
    out = {}
    kinds = {"mv": np.int32, "cost": np.int32, "sa8d": np.int32, "level": np.int16, "numSig": np.uint32, "dist": np.uint64, "clevel": np.int16,
             "cnumSig": np.uint32, "cdist": np.uint64, "mv1": np.int32, "cost1": np.int32}

    def put(name, a):
        dt = kinds.get(name.split("[")[0])
        a = np.ascontiguousarray(a if dt is None else np.asarray(a).astype(dt, copy=False))
        out[name] = hashlib.sha256(a.tobytes()).hexdigest()
    for k in ("mv", "cost", "sa8d", "level", "numSig", "dist", "clevel", "cnumSig", "cdist", "pred_c", "recon_c", "mv1", "cost1"):
        if k in res:
            for i, a in enumerate(res[k]):
                put("%s[%d]" % (k, i), a)
    put("pred", res["pred"])
    put("recon", res["recon"])
    return out


def run_oracle(name):
    from frame_oracle import oracle_frame_pass
    case = CASES[name]
    w, h, depth, qp, method, merange, subme, seed, kind = case
    sc, nxt = scene(case)
    kw = dict(depth=depth, qp=qp, merange=merange, method=method, subme=subme, src_c=(sc["src_cb"], sc["src_cr"]), ref_c=(sc["ref_cb"], sc["ref_cr"]))
    if nxt:
        kw.update(ref1=nxt[0], ref1_c=(nxt[1], nxt[2]))
    r = oracle_frame_pass(sc["src"], sc["ref"], **kw)
    return name, digest(r), {"nonzero_mvs": int(sum(int(np.any(m != 0, axis=1).sum()) for m in r["mv"])), "numSig": int(sum(int(x.sum()) for x in r["numSig"]))}


def main():
    out = {"generator": "tests/golden/make_framepass_golden.py", "cases": {}}
    with ProcessPoolExecutor(max_workers=min(len(CASES), os.cpu_count() or 1)) as ex:
        for name, dg, info in ex.map(run_oracle, list(CASES)):
            out["cases"][name] = {"params": dict(zip(("width", "height", "depth", "qp", "method", "merange", "subme", "seed", "pass"), CASES[name])),
                                  "digests": dg, "info": info}
            print(name, info, flush=True)
    with open(os.path.join(HERE, "framepass_configs_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
