#!/bin/bash
set -u
python3 -c "
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c.yuv', 1920, 1080, 60, seed=4321)"
R=$(pwd)/oracle/_ref
A="--input /tmp/c.yuv --input-res 1920x1080 --fps 30 --frames 60 --preset medium --me hex --hash 1"
TIMEFORMAT="hip wall %R s user %U s sys %S s"
{ time X265HIP_DEBUG_STARTUP=1 X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2> /tmp/st.log ; } 2>&1
grep "^encoded" /tmp/st.log
grep "x265hip-startup" /tmp/st.log | grep -v "source picture" > gpurun_out/r03_close_marks.txt
grep -c "PicYuv::destroy: retired" /tmp/st.log
