#!/bin/bash
# round 4, twelfth GPU call: whole-round launches of the SAD surfaces; the CU's final sse / psy out of the jobs (A/B); what creating per-picture device
# objects costs; can the host write a mailbox in device memory (large BAR) and what does the hand-off gain; bench line; CPU profile
set -u
OUT=gpurun_out/r04_l
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py tests/test_sadsurf.py tests/test_refpic.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 60 tools/micro/create_cost 2>&1 | tee $OUT/create_cost.txt
for k in 0 1 2 3; do timeout 30 tools/micro/bar_mailbox $k 2>&1 | tee $OUT/bar_mailbox_$k.txt; echo "exit status $?" | tee -a $OUT/bar_mailbox_$k.txt; done
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | cut -c1-700 | tee $OUT/verify.txt
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 on: dist1:X265HIP_CUSERVE_DIST=1 off:X265HIP_CUSERVE=0 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 600 $OUT/bench_line.json
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_CPUSAMPLE_OUT=/tmp/hip.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/prof/resolve.py /tmp/hip.bin 120 > $OUT/cpu_profile_hip.txt; head -30 $OUT/cpu_profile_hip.txt
