#!/bin/bash
set -u
OUT=gpurun_out/r03_f
mkdir -p $OUT
timeout 600 python -m pytest tests/test_places.py tests/test_sadsurf.py tests/test_refpic.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 600 python tools/ab_encode.py --rounds 2 --frames 120 base: "two:X265HIP_DEVICES=0,0" --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_f/ab.json"))
for k, v in d["configs"].items():
    print(k, [l for l in v["served"] if "places" in l or "device time" in l])
PY
