#!/bin/bash
# round 4, seventeenth GPU call: jobs submitted ahead at the skip evaluation (A/B), VERIFY
set -u
OUT=gpurun_out/r04_q
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
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 on: spec0:X265HIP_CUSERVE_SPEC=0 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_q/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:700] for l in v["served"] if "waits" in l or "ahead" in l])
PY
