#!/bin/bash
# round 4, twenty-second GPU call: how long the worker gathers attach jobs before it launches (30 / 100 / 300 / 1000 us): launches, device time, misses, fps
set -u
OUT=gpurun_out/r04_w
mkdir -p $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
timeout 900 python tools/ab_encode.py --rounds 2 --frames 120 g100: g30:X265HIP_SADSURF_GATHER_US=30 g300:X265HIP_SADSURF_GATHER_US=300 g1000:X265HIP_SADSURF_GATHER_US=1000 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_w/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[60:330] for l in v["served"] if "integer-pel SADs of" in l])
PY
