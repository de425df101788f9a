#!/bin/bash
# (experiment of round 3: an opt-in _exit() at the end of the bindings' exit handlers; wall - CLI time 0.55 s either way, so the switch was removed again —
# the script stays as the record of how profiles/r03_v5_exit_experiment.txt was produced)
set -u
python3 -c "
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c.yuv', 1920, 1080, 120, seed=4321)"
R=$(pwd)/oracle/_ref
A="--input /tmp/c.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
$R/x265_8bit $A -o /tmp/ref.hevc > /dev/null 2>&1
for i in 1 2 3; do
 for fe in 0 1; do
  TIMEFORMAT="fast_exit=$fe wall %R s user %U s sys %S s"
  { time X265HIP_FAST_EXIT=$fe X265HIP=require $R/x265_hip_8bit $A -o /tmp/a.hevc 2> /tmp/st.log ; } 2>&1
  grep "^encoded" /tmp/st.log | cut -c1-45
  cmp -s /tmp/ref.hevc /tmp/a.hevc && echo identical || echo DIFFERENT
 done
done
