"""Randomised parity sweep of the frame pass on a GPU box: picture sizes (any multiple of 8, including ones smaller than a CTU row or
column), depth, search method, subme, qp, merange, scene statistics and the pass flavour (luma / 4:2:0 / B) are drawn at random and every
output is compared bit for bit with the C restatement (the same checker tests/test_framepass.py uses).  Not part of the pytest suites —
a soak tool (it lives in tests/ because it uses the oracle):      python tests/fuzz_parity.py --cases 120 --seed 1
Prints one line per failing case and a JSON summary; exit code 1 on any mismatch."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from frame_oracle import make_scene_yuv, oracle_frame_pass, same_results      # noqa: E402
from x265_amd import hipprim as hp, framepass                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-pixels", type=int, default=400 * 300)
    a = ap.parse_args()
    hp.check(hp.lib().x265hip_init(0))
    rng = np.random.default_rng(a.seed)
    bad, done, t0 = [], 0, time.time()
    for c in range(a.cases):
        while True:
            w, h = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 40)) * 8
            if w * h <= a.max_pixels:
                break
        depth = int(rng.choice([8, 8, 10, 12]))
        method = int(rng.choice([0, 1, 1, 2, 3]))
        subme = int(rng.integers(0, 8))
        qp = int(rng.integers(4, 50))
        merange = int(rng.choice([8, 16, 25, 57]))
        flavour = str(rng.choice(["luma", "yuv", "yuv", "b"]))
        sc_kw = dict(depth=depth, seed=int(rng.integers(0, 1 << 30)), tile=int(rng.choice([16, 32, 48, 96])), vmax=int(rng.choice([0, 2, 9, 20])),
                     sigma=float(rng.choice([0.0, 1.0, 3.0, 12.0])) * (1 << (depth - 8)))
        label = dict(w=w, h=h, depth=depth, method=method, subme=subme, qp=qp, merange=merange, flavour=flavour, scene=sc_kw)
        try:
            sc = make_scene_yuv(w, h, **sc_kw)
            fp = framepass.FramePass(w, h, depth=depth, qp=qp, merange=merange, method=method, subme=subme)
            if flavour == "luma":
                got = fp.run_host(sc["src"], sc["ref"])
                want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, merange=merange, method=method, subme=subme)
            elif flavour == "yuv":
                got = fp.run_host_yuv(sc)
                want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, merange=merange, method=method, subme=subme,
                                         src_c=(sc["src_cb"], sc["src_cr"]), ref_c=(sc["ref_cb"], sc["ref_cr"]))
            else:
                kw2 = dict(sc_kw)
                kw2["seed"] = sc_kw["seed"] ^ 0x5a5a
                nxt = make_scene_yuv(w, h, **kw2)
                r1 = (nxt["src"], nxt["src_cb"], nxt["src_cr"])
                got = fp.run_host_yuv_b(sc, *r1)
                want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, merange=merange, method=method, subme=subme,
                                         src_c=(sc["src_cb"], sc["src_cr"]), ref_c=(sc["ref_cb"], sc["ref_cr"]), ref1=r1[0], ref1_c=(r1[1], r1[2]))
            diff = same_results(got, want)
            del fp
        except Exception as e:                                    # noqa: BLE001  (a soak tool reports and goes on)
            diff = ["exception: %r" % (e,)]
        done += 1
        if diff:
            bad.append(dict(case=c, cfg=label, diff=[str(d) for d in diff[:6]]))
            print("MISMATCH", json.dumps(bad[-1]), flush=True)
    print(json.dumps(dict(cases=done, mismatches=len(bad), seed=a.seed, seconds=round(time.time() - t0, 1))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
