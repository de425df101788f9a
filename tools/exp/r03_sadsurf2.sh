#!/bin/bash
# Round 3: why do the SAD lookups not show in the fps?  CPU time and wall of the 1080p encode in several configurations, + the pinned-memory probe
set -u
OUT=gpurun_out/r03_ss2
mkdir -p $OUT
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c1080.yuv', 1920, 1080, 120, seed=4321)
PY
R=oracle/_ref
A="--input /tmp/c1080.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
run() { tag=$1; shift
    for i in 1 2 3; do
        TIMEFORMAT="$tag wall %R s user %U s sys %S s"
        { time env "$@" X265HIP=require X265HIP_VERBOSE=1 timeout 300 $R/x265_hip_8bit $A -o /tmp/$tag.hevc 2> $OUT/$tag.$i.log ; } 2>&1
        grep -h "^encoded\|x265hip: sadplanes" $OUT/$tag.$i.log
    done
}
run sad0 X265HIP_SADPLANES=0
run sad1 X265HIP_SADPLANES=1
run sad1_l12 X265HIP_SADPLANES=1 X265HIP_SADPLANES_LEVELS=12
run sad1_l8 X265HIP_SADPLANES=1 X265HIP_SADPLANES_LEVELS=8
run time_on X265HIP_SADPLANES=1 X265HIP_DEBUG_SADTIME=1
run time_off X265HIP_SADPLANES=1 X265HIP_DEBUG_SADTIME=2
