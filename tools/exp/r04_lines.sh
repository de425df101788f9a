#!/bin/bash
# round 4: the mailbox as tagged 64-byte lines that the server's poll reads whole (no "see the doorbell, then fetch" round trip)
set -u
OUT=gpurun_out/r04_lines
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/pytest.txt
X265HIP_CUSERVE_MAILBOX=host timeout 300 python -m pytest tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -2 | tee -a $OUT/pytest.txt
timeout 100 tools/micro/cuserve_rt 0 2000 1 2>&1 | cut -c1-400 | tee $OUT/cuserve_rt_stamps.txt | head -9
for m in device host; do X265HIP_CUSERVE_MAILBOX=$m timeout 100 tools/micro/cuserve_rt 0 3000 2>&1 | cut -c1-300 > $OUT/cuserve_rt_mailbox_$m.txt; done
timeout 100 tools/micro/cuserve_rt 1 1000 2>&1 | cut -c1-300 > $OUT/cuserve_rt_mode1.txt; head -2 $OUT/cuserve_rt_mode1.txt | cut -c1-200
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
ARGS="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --input-depth 8 --fps 30 --frames 60 --preset medium --hash 1 --me hex"
X265HIP=require X265HIP_VERBOSE=1 X265HIP_VERIFY=1 timeout 300 oracle/_ref/x265_hip_8bit $ARGS -o /tmp/verify.hevc 2>&1 | grep -v "^\[" | grep "cuserve: [0-9]* CU\|VERIFY\|encoded" | cut -c1-300 | tee $OUT/verify.txt
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 on: hostbox:X265HIP_CUSERVE_MAILBOX=host --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_lines/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[l.find("waits") - 8:][:420] for l in v["served"] if "waits by" in l or "waits of" in l])
PY
