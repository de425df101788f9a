#!/bin/bash
# round 4, first GPU call: "measure first" for the CU-level offload
#   (a) the hand-off: launch + sync / launch + host flag / resident-kernel mailbox, 1-16 host threads (tools/micro/handoff)
#   (b) the prize: cycles inside Quant::transformNxN / ::invtransformNxN / Search::estimateResidualQT / ::checkIntraInInter by block size in the
#       bound encoder at BASELINE configs[1] (X265HIP_DEBUG_CUTIME=1; x265_amd/host/x265_hip_cuserve.cpp)
set -u
OUT=gpurun_out/r04_a
mkdir -p $OUT
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null; lscpu | grep -i "model name" >> $OUT/host.txt
timeout 300 tools/micro/handoff 3000 2>&1 | tee $OUT/handoff.txt
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 120 --preset medium --hash 1 --me hex"
( time X265HIP=require X265HIP_VERBOSE=1 X265HIP_DEBUG_CUTIME=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/cutime.hevc ) 2>&1 | grep -v "^\[" | tail -45 | tee $OUT/cutime_1080p_medium.txt
( time X265HIP=require X265HIP_VERBOSE=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/plain.hevc ) 2>&1 | grep -v "^\[" | tail -12 | tee $OUT/plain_1080p_medium.txt
cmp /tmp/cutime.hevc /tmp/plain.hevc && echo "bitstreams identical" | tee -a $OUT/plain_1080p_medium.txt
