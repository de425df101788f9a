#!/bin/bash
# Round 3: where the bound encoder's CPU time goes on the MI355X box (SIGPROF sampling, tools/prof), default pools
set -u
OUT=gpurun_out/r03_e
mkdir -p $OUT
python3 -c "
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c.yuv', 1920, 1080, 120, seed=4321)"
R=$(pwd)/oracle/_ref
A="--input /tmp/c.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_CPUSAMPLE_OUT=/tmp/hip.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/prof/resolve.py /tmp/hip.bin 90 | tee $OUT/cpu_profile_hip.txt | head -40
X265HIP_CPUSAMPLE_OUT=/tmp/ref.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so $R/x265_8bit $A -o /tmp/r.hevc 2>&1 | grep "^encoded"
python3 tools/prof/resolve.py /tmp/ref.bin 60 > $OUT/cpu_profile_ref.txt
