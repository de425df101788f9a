#!/bin/bash
# round 4, sixth GPU call: who calls the ioctls (CPU sampler with call chains); sub-pel hit rate of the SAD surfaces' best vectors on the real path;
# A/B of the lookahead staging
set -u
OUT=gpurun_out/r04_f
mkdir -p $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_CPUSAMPLE_US=5000 X265HIP_CPUSAMPLE_STACK=1 X265HIP_CPUSAMPLE_OUT=/tmp/hip.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/prof/callers.py /tmp/hip.bin 'ioctl|munmap|mmap' 14 | cut -c1-700 | tee $OUT/ioctl_callers.txt
python3 tools/prof/callers.py /tmp/hip.bin 'libhsa|libamdhip|nss_database|comgr' 10 | cut -c1-600 | tee $OUT/runtime_callers.txt
X265HIP=require X265HIP_VERBOSE=1 X265HIP_DEBUG_SUBPELHIT=1 $R/x265_hip_8bit $A -o /tmp/b.hevc 2>&1 | grep "quarter-pels\|^encoded" | tee $OUT/subpel_hit.txt
timeout 600 python tools/ab_encode.py --rounds 3 --frames 120 off:X265HIP_CUSERVE=0 on: --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
