#!/bin/bash
# round 4, thirteenth GPU call: the job mailbox's host -> device half in device memory (large BAR) vs in host memory; sub_ps / add_ps calls nobody reads
# put off; chroma reconstruction answers; stream pool for source-picture uploads (start-up marks)
set -u
OUT=gpurun_out/r04_m
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py tests/test_sadsurf.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
X265HIP_CUSERVE_MAILBOX=host timeout 300 python -m pytest tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -2 | tee -a $OUT/pytest.txt
for m in device host; do X265HIP_CUSERVE_MAILBOX=$m timeout 120 tools/micro/cuserve_rt 0 3000 2>&1 | tee $OUT/cuserve_rt_mailbox_$m.txt | tail -8; done
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | cut -c1-700 | tee $OUT/verify.txt
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 on: hostbox:X265HIP_CUSERVE_MAILBOX=host dist2:X265HIP_CUSERVE_DIST=2 off:X265HIP_CUSERVE=0 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_m/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:600] for l in v["served"] if "cuserve" in l])
PY
X265HIP=require X265HIP_VERBOSE=1 X265HIP_DEBUG_STARTUP=1 oracle/_ref/x265_hip_8bit --input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex -o /dev/null 2>&1 | grep "startup\|^encoded" | grep -v "PicYuv::destroy" | tee $OUT/startup_marks.txt | tail -5
