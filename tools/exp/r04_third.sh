#!/bin/bash
# round 4, third GPU call: the CU jobs after the latency work (one round trip for header + pixels, forward half published first, operands built once,
# slots taken per job), parity in both hand-off modes, round trip, encode A/B
set -u
OUT=gpurun_out/r04_c
mkdir -p $OUT
timeout 60 tools/micro/mailbox_diag 2>&1 | tail -2 | tee $OUT/mailbox_diag_memory.txt
timeout 400 python -m pytest "tests/test_cuserve.py::test_device_jobs_match_the_restatement[1]" -x -q 2>&1 | tail -12 | tee $OUT/pytest_mode1.txt
timeout 200 python -m pytest "tests/test_cuserve.py::test_device_jobs_match_the_restatement[0]" -x -q 2>&1 | tail -12 | tee $OUT/pytest_mode0.txt
timeout 120 tools/micro/cuserve_rt 1 2000 2>&1 | tee $OUT/cuserve_rt_mode1.txt
timeout 120 tools/micro/cuserve_rt 0 2000 2>&1 | tee $OUT/cuserve_rt_mode0.txt
CFG="off:X265HIP_CUSERVE=0 r32:X265HIP_CUSERVE_MODE=0,X265HIP_CUSERVE_MIN=32 r64:X265HIP_CUSERVE_MODE=0,X265HIP_CUSERVE_MIN=64 r32i:X265HIP_CUSERVE_MODE=0,X265HIP_CUSERVE_MIN=32,X265HIP_CUSERVE_IDLE_US=20000 r16:X265HIP_CUSERVE_MODE=0,X265HIP_CUSERVE_MIN=16"
timeout 1100 python tools/ab_encode.py --rounds 3 --frames 120 $CFG --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_c/ab.json"))
for k, v in d["configs"].items():
    print(k, [l for l in v["served"] if "cuserve" in l or "device time" in l])
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 40 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 X265HIP_CUSERVE_MODE=0 X265HIP_CUSERVE_MIN=16 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | tee $OUT/verify.txt
