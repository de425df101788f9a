#!/bin/bash
# round 4, twentieth GPU call: fewer resident server workgroups (2 x CPUs = 32 here instead of 64): CUs not submitted, fps, the SAD surfaces' live VALU fraction
set -u
OUT=gpurun_out/r04_t
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py tests/test_sadsurf.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/pytest.txt
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd.synth import make_clip
make_clip("/tmp/ab_clip_1920x1080_120.yuv", 1920, 1080, 120, seed=4321)
PY
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 s32: s64:X265HIP_CUSERVE_SLOTS=64 s20:X265HIP_CUSERVE_SLOTS=20 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_t/ab1080.json"))
for k, v in d["configs"].items():
    print(k, [l[:120] + " ... " + l[l.find("waits") - 12:][:80] for l in v["served"] if "handed to the GPU" in l], [l[60:300] for l in v["served"] if "integer-pel SADs of" in l])
PY
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 200 $OUT/bench_line.json
