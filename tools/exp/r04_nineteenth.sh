#!/bin/bash
# round 4, nineteenth GPU call: how long SAD-surface rows may wait for company (deferral limit 2 / 8 / 16 bands): misses, fps, live VALU fraction
set -u
OUT=gpurun_out/r04_s
mkdir -p $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
timeout 900 python tools/ab_encode.py --rounds 2 --frames 120 d8: d2:X265HIP_SADSURF_DEFER=2 d16:X265HIP_SADSURF_DEFER=16 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:420] for l in v["served"] if "sadplanes: " in l and "integer-pel SADs of" in l])
PY
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 200 $OUT/bench_line.json
