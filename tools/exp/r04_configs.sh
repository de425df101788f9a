#!/bin/bash
# round 4, final tree: BASELINE configs[2] (4K slow, star) and configs[3] (4K Main10 slower --rd 6) against round 3's feature set and the reference
set -u
OUT=gpurun_out/r04_c23
mkdir -p $OUT
R3="X265HIP_CUSERVE=0,X265HIP_SADPLANES_SUBPEL=0,X265HIP_SADPLANES_RECT=0,X265HIP_SADSURF_BATCH=0,X265HIP_SADSURF_ROUNDS=0,X265HIP_SADSURF_GATHER_US=0,X265HIP_PINNED=hip"
timeout 900 python tools/ab_encode.py --rounds 2 --frames 24 --res 3840x2160 --preset slow --extra "--me star --merange 57" base: r3:$R3 --out $OUT/configs2.json 2>&1 | tee $OUT/configs2_4k_slow_star_ab.txt
timeout 900 python tools/ab_encode.py --rounds 2 --frames 8 --res 3840x2160 --preset slower --extra "--rd 6" --bits 10 base: r3:$R3 --out $OUT/configs3.json 2>&1 | tee $OUT/configs3_4k_main10_slower_ab.txt
python3 - <<'PY'
import json
for f in ("configs2", "configs3"):
    d = json.load(open("gpurun_out/r04_c23/%s.json" % f))
    for l in d["configs"]["base"]["served"]:
        if "sadplanes" in l: print(f, l[:330])
PY
