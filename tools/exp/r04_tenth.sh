#!/bin/bash
# round 4, tenth GPU call: page-locking with transparent huge pages; jobs submitted at encodeResAndCalcRdInterCU entry (A/B); configs[3] again with the rebuilt Main10 binary
set -u
OUT=gpurun_out/r04_j
mkdir -p $OUT
timeout 120 tools/micro/pin_thp 2>&1 | tee $OUT/pin_thp.txt
R3="X265HIP_SADPLANES_RECT=0,X265HIP_SADPLANES_SUBPEL=0,X265HIP_CUSERVE=0"
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 r3:$R3 on: --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_j/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:500] for l in v["served"] if "cuserve" in l])
PY
timeout 900 python tools/ab_encode.py --rounds 2 --frames 8 --res 3840x2160 --preset slower --extra "--rd 6" --bits 10 base: r3:$R3 --out $OUT/configs3.json 2>&1 | tee $OUT/configs3_4k_main10_slower_ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_j/configs3.json"))
for k, v in d["configs"].items():
    print(k, [l[:300] for l in v["served"] if "rectangular" in l or "sub-pel SATDs" in l])
PY
X265HIP=require X265HIP_VERBOSE=1 X265HIP_DEBUG_STARTUP=1 oracle/_ref/x265_hip_8bit --input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex -o /dev/null 2>&1 | grep "startup\|^encoded" | tee $OUT/startup_marks.txt
