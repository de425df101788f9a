"""Count, with the pinned CPU oracle, the per-call ALGORITHMIC traffic (SURVEY.md §8d figures: sad 2WHB, sad_x3 4WHB, sad_x4 5WHB,
an interpolated candidate adds the filter's input + output) that the REFERENCE's motion search issues per PU on bench.py's
workload (1920x1080 8-bit, seed 4321, HEX, merange 57, subme 2).  The search is bit-exact between oracle and GPU, so the
candidate sequence — and therefore this count — is the same on both.  bench.py's ME_BYTES_PER_PU comes from this tool:

    python tools/count_me_units.py          # prints the dict to paste
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from frame_oracle import oracle_frame_pass
    from oracle import pyoracle as po
    from x265_amd.synth import make_scene_yuv
    W, H, depth = 1920, 1080, 8
    sc = make_scene_yuv(W, H, depth=depth, seed=4321)
    oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=28, merange=57, method=1, subme=2,
                      src_c=(sc["src_cb"], sc["src_cr"]), ref_c=(sc["ref_cb"], sc["ref_cr"]))
    L = po.oracle()
    out = (C.c_uint64 * 12)()
    L.orc_frame_pass_me_stats.restype = None
    L.orc_frame_pass_me_stats(out)
    res = {}
    for l, size in enumerate((64, 32, 16, 8)):
        byts, calls, pus = int(out[3 * l]), int(out[3 * l + 1]), int(out[3 * l + 2])
        res["me%d" % size] = {"pus": pus, "bytes_per_pu": round(byts / pus, 1), "calls_per_pu": round(calls / pus, 2), "bytes": byts}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
