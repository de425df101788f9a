#!/bin/bash
# round 4, fourth GPU call: the whole GPU test tier, the encode A/B with the jobs' distortion / psy answers, the bench line, rocprofv3 evidence
set -u
OUT=gpurun_out/r04_d
mkdir -p $OUT
( time timeout 1200 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -12 | tee $OUT/gpu_test_tier.txt
timeout 120 tools/micro/cuserve_rt 0 2000 2>&1 | tee $OUT/cuserve_rt_mode0.txt
CFG="off:X265HIP_CUSERVE=0 on: tr:X265HIP_CUSERVE_DIST=0 on64:X265HIP_CUSERVE_MIN=64"
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 $CFG --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_d/ab.json"))
for k, v in d["configs"].items():
    print(k, [l for l in v["served"] if "cuserve" in l or "device time" in l])
PY
( time timeout 600 python bench.py ) 2>&1 | tail -4 | tee $OUT/bench.txt
bash tools/collect_profiles_r04.sh r04_v1 2>&1 | tail -30 | tee $OUT/collect.txt
