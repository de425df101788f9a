#!/bin/bash
# round 4, fifth GPU call: the job server after the start / leave race fixes and without the L2-invalidating fences; where the device-side time goes;
# parity again; VERIFY encode; A/B incl. the surface-row batching; the bench line; a CPU profile of the bound encoder
set -u
OUT=gpurun_out/r04_e
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py tests/test_sadsurf.py tests/test_places.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.txt
for i in 1 2 3; do timeout 120 tools/micro/cuserve_rt 0 3000 2>&1 | tee $OUT/cuserve_rt_mode0_$i.txt | tail -8; done
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve\|VERIFY\|encoded" | tee $OUT/verify.txt
CFG="off:X265HIP_CUSERVE=0 on: nobatch:X265HIP_SADSURF_BATCH=0 idle10:X265HIP_CUSERVE_IDLE_US=10000"
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 $CFG --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_e/ab.json"))
for k, v in d["configs"].items():
    print(k, [l for l in v["served"] if "cuserve" in l or "device time" in l or "sadplanes" in l])
PY
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 1500 $OUT/bench_line.json
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_CPUSAMPLE_OUT=/tmp/hip.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/prof/resolve.py /tmp/hip.bin 90 > $OUT/cpu_profile_hip.txt; head -60 $OUT/cpu_profile_hip.txt
