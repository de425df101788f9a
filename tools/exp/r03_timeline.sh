#!/bin/bash
set -u
OUT=gpurun_out/r03_tl
mkdir -p $OUT
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c1080.yuv', 1920, 1080, 120, seed=4321)
PY
R=oracle/_ref
A="--input /tmp/c1080.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_DEBUG_LA_TIMELINE=/tmp/tl.txt X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/la_timeline.py /tmp/tl.txt | tee $OUT/timeline_256.txt
X265HIP_DEBUG_LA_TIMELINE=/tmp/tl2.txt X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A --pools 32 -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/la_timeline.py /tmp/tl2.txt | tee $OUT/timeline_32.txt
