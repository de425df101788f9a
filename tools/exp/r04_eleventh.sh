#!/bin/bash
# round 4, eleventh GPU call: coded-side psy energy in the jobs, huge-page page-locking — parity, VERIFY, A/B, wall clock of a whole run
set -u
OUT=gpurun_out/r04_k
mkdir -p $OUT
timeout 600 python -m pytest tests/test_cuserve.py tests/test_refpic.py tests/test_sadsurf.py tests/test_places.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | cut -c1-700 | tee $OUT/verify.txt
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 on: hippin:X265HIP_PINNED=hip off:X265HIP_CUSERVE=0 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_k/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:500] for l in v["served"] if "cuserve" in l])
PY
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex -o /dev/null"
for v in thp hip; do for i in 1 2 3; do /usr/bin/time -f "$v wall %e s user %U sys %S" env X265HIP=require X265HIP_PINNED=$v oracle/_ref/x265_hip_8bit $A 2>&1 | grep "wall\|^encoded"; done; done | tee $OUT/wall.txt
for i in 1 2; do /usr/bin/time -f "reference wall %e s user %U sys %S" oracle/_ref/x265_8bit $A 2>&1 | grep "wall\|^encoded"; done | tee -a $OUT/wall.txt
