#!/bin/bash
# round 4, ninth GPU call: the other BASELINE shapes (configs[2] 4K slow star, configs[3] 4K Main10 slower rd 6) with and without this round's lookups
# (rectangular PUs, sub-pel SATD tables), and the 1080p A/B of everything
set -u
OUT=gpurun_out/r04_i
mkdir -p $OUT
R3="X265HIP_SADPLANES_RECT=0,X265HIP_SADPLANES_SUBPEL=0,X265HIP_CUSERVE=0"
timeout 900 python tools/ab_encode.py --rounds 2 --frames 24 --res 3840x2160 --preset slow --extra "--me star --merange 57" base: r3:$R3 --out $OUT/configs2.json 2>&1 | tee $OUT/configs2_4k_slow_star_ab.txt
timeout 900 python tools/ab_encode.py --rounds 2 --frames 8 --res 3840x2160 --preset slower --extra "--rd 6" --bits 10 base: r3:$R3 --out $OUT/configs3.json 2>&1 | tee $OUT/configs3_4k_main10_slower_ab.txt
python3 - <<'PY'
import json
for f in ("configs2", "configs3"):
    d = json.load(open("gpurun_out/r04_i/%s.json" % f))
    for k, v in d["configs"].items():
        print(f, k, [l[:400] for l in v["served"] if "rectangular" in l or "sub-pel SATDs" in l or "cuserve: " in l])
PY
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 r3:$R3 on: --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
