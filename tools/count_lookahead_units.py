"""Count, with the pinned CPU oracle, the per-call ALGORITHMIC traffic (sad / satd 2WHB; a quarter-pel candidate adds the two
half-pel blocks read and the averaged block written by pixelavg_pp) that the REFERENCE's lookahead issues per 8x8 lowres block in one
P-frame cost pass (CostEstimateGroup::estimateCUCost, slicetype.cpp:3218) on bench.py's probe geometry: 1920x1080 seed 4321 -> 960x544 lowres,
one slice.  The pass is bit-exact between oracle and GPU, so the candidate sequence — and this count — is the same on both.

    python tools/count_lookahead_units.py          # prints bytes / calls per block for bench.py's LA_BYTES_PER_BLOCK"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from backends import Orc
    from oracle import pyoracle as po
    from x265_amd.synth import make_scene
    W, H, depth, M = 1920, 1080, 8, 96
    sc = make_scene(W, H, depth, seed=4321)
    pad = lambda a: np.ascontiguousarray(np.pad(a, ((M, M), (M, M + 8)), mode="edge"))  # noqa: E731
    o = Orc(depth)
    L = po.oracle()
    L.orc_me_stats_reset.restype = None
    L.orc_me_stats.restype = None
    hcu = ((H // 2 + 7) // 8)
    L.orc_me_stats_reset()
    r = o.lookahead_cost_p(pad(sc["ref"]), pad(sc["src"]), (M, M), W, H, 32, 32, hcu, 1)
    out = (C.c_uint64 * 2)()
    L.orc_me_stats(out)
    ncu = len(r[2])
    print(json.dumps({"blocks": ncu, "bytes_per_block": round(out[0] / ncu, 1), "calls_per_block": round(out[1] / ncu, 2), "costEst": int(r[0])}))


if __name__ == "__main__":
    main()
