#!/bin/bash
# Round-3 "measure first" experiments on the MI355X box (VERDICT r02 task 1 / 2):
#   * issue rate of the SAD instructions (tools/micro/qsad_rate)
#   * the integer-pel SADs of motionEstimate that a surface lookup could serve: how many, how far from a centre (X265HIP_DEBUG_SADEXP=1),
#     and what they cost on the encode's critical path (X265HIP_DEBUG_SADEXP=2: every eligible call computed twice)
#   * the lookahead seam's batch composition (X265HIP_DEBUG_TRACE=1)
set -u
OUT=gpurun_out/r03_exp
mkdir -p $OUT
tools/micro/qsad_rate > $OUT/qsad_rate.txt 2>&1
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c1080.yuv', 1920, 1080, 120, seed=4321)
PY
R=oracle/_ref
A="--input /tmp/c1080.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
run() { # tag, env...
    tag=$1; shift
    for i in 1 2; do
        env "$@" X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/$tag.hevc 2> $OUT/$tag.$i.log
        grep -h "^encoded\|x265hip:" $OUT/$tag.$i.log | head -20
    done
}
$R/x265_8bit $A -o /tmp/ref.hevc 2>&1 | grep "^encoded" | tee $OUT/ref.log
run base X265HIP_NOP=1
run sadexp1 X265HIP_DEBUG_SADEXP=1
run sadexp2 X265HIP_DEBUG_SADEXP=2
run trace X265HIP_DEBUG_TRACE=1
for t in base sadexp1 sadexp2 trace; do cmp /tmp/ref.hevc /tmp/$t.hevc && echo "$t identical"; done | tee $OUT/identity.txt
grep -c "x265hip-trace" $OUT/trace.1.log
grep "x265hip-trace" $OUT/trace.1.log | awk '{print $3,$5,$7,$9,$11,$13}' > $OUT/trace_compact.txt
rm -f $OUT/trace.1.log $OUT/trace.2.log
