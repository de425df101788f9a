"""Randomised parity sweep of the lookahead front end on a GPU box (a soak tool like tests/fuzz_parity.py; not collected by pytest): picture
sizes (full resolution, any multiple of 16 from 32 up), depth, rows per slice / slice count and the B flavour at random, lowres init +
intra estimate + P / B cost passes, HIP vs the oracle.      python tests/fuzz_lookahead.py --cases 60"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backends import Orc, same                              # noqa: E402
from cases import lookahead_scene, lookahead_scene3        # noqa: E402
import hipbackend                                           # noqa: E402
from x265_amd import hipprim as hp                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    hp.check(hp.lib().x265hip_init(0))
    rng = np.random.default_rng(a.seed)
    bad, t0 = [], time.time()
    for c in range(a.cases):
        depth = int(rng.choice([8, 10, 12]))
        w, h = int(rng.integers(2, 60)) * 16, int(rng.integers(2, 40)) * 16
        hcu = (h // 2 + 7) // 8                             # lowres height in 8x8 blocks
        ns = min(int(rng.integers(1, 4)), hcu)
        rps = (hcu + ns - 1) // ns
        while ns > 1 and (ns - 1) * rps >= hcu:             # every slice owns at least one block row (as the lookahead's own split does)
            ns -= 1
            rps = (hcu + ns - 1) // ns
        b = int(rng.integers(0, 2))
        o, g = Orc(depth), hipbackend.Hip(depth)
        cfg = dict(w=w, h=h, depth=depth, rps=rps, ns=ns, b=b)
        try:
            if b:
                pics, m = lookahead_scene3(depth, int(rng.integers(0, 1 << 20)), h, w)
                pre = int(rng.integers(0, 2))
                want = o.lookahead_cost_b(pics[0], pics[1], pics[2], (m, m), w, h, m, m, rps, ns, pre)
                got = g.lookahead_cost_b(pics[0], pics[1], pics[2], (m, m), w, h, m, m, rps, ns, pre)
            else:
                s0, s1, m = lookahead_scene(depth, int(rng.integers(0, 1 << 20)), h, w)
                want = o.lookahead_cost_p(s0, s1, (m, m), w, h, m, m, rps, ns)
                got = g.lookahead_cost_p(s0, s1, (m, m), w, h, m, m, rps, ns)
            hipbackend._release()
            ok = all(same(x, y) for x, y in zip(want, got)) and len(want) == len(got)
        except Exception as e:                              # noqa: BLE001
            ok = False
            cfg["exception"] = repr(e)
        if not ok:
            bad.append(cfg)
            print("MISMATCH", json.dumps(cfg), flush=True)
    print(json.dumps(dict(cases=a.cases, mismatches=len(bad), seconds=round(time.time() - t0, 1))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
