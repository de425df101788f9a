#!/bin/bash
# round 4, seventh GPU call: sub-pel SATD tables — device parity, VERIFY encode, A/B
set -u
OUT=gpurun_out/r04_g
mkdir -p $OUT
timeout 600 python -m pytest tests/test_sadsurf.py tests/test_refpic.py tests/test_places.py tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.txt
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded\|sub-pel\|device time" | cut -c1-1200 | tee $OUT/verify.txt
CFG="off:X265HIP_CUSERVE=0,X265HIP_SADPLANES_SUBPEL=0 cu:X265HIP_SADPLANES_SUBPEL=0 on: sp:X265HIP_CUSERVE=0"
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 $CFG --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_g/ab.json"))
for k, v in d["configs"].items():
    print(k, [l[:600] for l in v["served"] if "sub-pel" in l or "device time" in l])
PY
