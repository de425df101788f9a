"""Randomised parity sweep of the primitives and of motionEstimate on a GPU box (a soak tool next to tests/fuzz_parity.py; not collected by
pytest): the TestBench-shaped case generator of tests/cases.py with other seeds, and motionEstimate batches over every PU shape, search
method, subme and candidate count on differently-seeded scenes, HIP vs the oracle.      python tests/fuzz_primitives.py --seeds 3"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backends import Orc, PU_SIZES, same          # noqa: E402
from cases import gen_cases, me_scene, coef_cases, loop_cases, deblock_cases, cutree_cases             # noqa: E402
import hipbackend                                 # noqa: E402
from x265_amd import hipprim as hp                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--first-seed", type=int, default=100)
    a = ap.parse_args()
    hp.check(hp.lib().x265hip_init(0))
    bad, n, t0 = [], 0, time.time()
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        # the "next" rows: coefficient-scan helpers, in-loop filter primitives, deblocking edges, CU-tree step
        o8, g8 = Orc(8), hipbackend.Hip(8)
        for label, fn, args in coef_cases(seed=seed):
            n += 1
            if not same(getattr(o8, fn)(*args), getattr(g8, fn)(*args)):
                bad.append("seed %d %s" % (seed, label))
            hipbackend._release()
        for label, args in cutree_cases(seed=seed):
            w_, g_ = o8.cutree_propagate(*args), g8.cutree_propagate(*args)
            n += 1
            if not (np.array_equal(w_[0], g_[0]) and np.array_equal(w_[1], g_[1])):
                bad.append("seed %d %s" % (seed, label))
            hipbackend._release()
        for depth in (8, 10, 12):
            o, g = Orc(depth), hipbackend.Hip(depth)
            import itertools
            for label, fn, args in itertools.chain(loop_cases(depth, seed=seed), deblock_cases(depth, seed=seed)):
                n += 1
                if not same(getattr(o, fn)(*args), getattr(g, fn)(*args)):
                    bad.append("seed %d depth %d %s" % (seed, depth, label))
                hipbackend._release()
        for depth in (8, 10, 12):
            o, g = Orc(depth), hipbackend.Hip(depth)
            if depth != 12:
                for label, fn, args in gen_cases(depth, seed=seed, reps=1):
                    want = getattr(o, fn)(*args)
                    got = getattr(g, fn)(*args)
                    hipbackend._release()
                    n += 1
                    if not same(got, want):
                        bad.append("seed %d depth %d %s" % (seed, depth, label))
            rng = np.random.default_rng(seed * 7 + depth)
            refp, srcp, m = me_scene(depth, seed * 13 + depth)
            H, W = refp.shape[0] - 2 * m, refp.shape[1] - 2 * m
            for qp in (22, 37):
                g.set_mvcost_table(qp, o.mvcost_table(qp))
            for (w, h) in PU_SIZES:
                if (w, h) == (4, 4):
                    continue
                for method in (0, 1, 2, 3):
                    subme, qp, numCand = int(rng.integers(0, 8)), int(rng.choice([22, 37])), int(rng.integers(0, 4))
                    merange = int(rng.choice([8, 16, 32, 57]))
                    planes = int(rng.integers(0, 2))
                    pus, mins, maxs, mvps, cands = [], [], [], [], []
                    for _ in range(10):
                        bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                        by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                        qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                        mvmin = [(qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange]
                        mvmax = [(qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange]
                        k = int(rng.integers(0, 4))
                        if k == 0:
                            mvmax[1] = max(min(mvmax[1], int(rng.integers(0, 6))), mvmin[1])
                        if k == 1:
                            mvmin[1] = min(max(mvmin[1], int(rng.integers(-3, 4))), mvmax[1])
                        pus.append((bx, by)); mins.append(tuple(mvmin)); maxs.append(tuple(mvmax)); mvps.append(qmvp)
                        cands.append([(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(numCand)])
                    cost, mv = g.motion_estimate_batch(refp, srcp, w, h, pus, mins, maxs, mvps, cands if numCand else [], merange, method, subme, qp,
                                                       planes_margin=m if planes else 0)
                    hipbackend._release()
                    for i in range(len(pus)):
                        want = o.motion_estimate(refp, srcp, pus[i][0], pus[i][1], w, h, mins[i], maxs[i], mvps[i], cands[i], merange, method, subme, qp)
                        n += 1
                        if (int(cost[i]), (int(mv[i, 0]), int(mv[i, 1]))) != want:
                            bad.append("seed %d depth %d me%d %dx%d subme%d merange %d planes %d got %s want %s"
                                       % (seed, depth, method, w, h, subme, merange, planes, (int(cost[i]), tuple(int(v) for v in mv[i])), want))
    for b in bad[:20]:
        print("MISMATCH", b)
    print(json.dumps(dict(checks=n, mismatches=len(bad), seconds=round(time.time() - t0, 1))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
