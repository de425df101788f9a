#!/bin/bash
# round 4, sixteenth GPU call: where the host waits (by site); all-zero tiles skip their inverse half; 16x16 CUs as jobs again now that the hand-off is shorter
set -u
OUT=gpurun_out/r04_p
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/pytest.txt
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | cut -c1-700 | tee $OUT/verify.txt
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 on: min16:X265HIP_CUSERVE_MIN=16 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_p/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:700] for l in v["served"] if "cuserve" in l])
PY
