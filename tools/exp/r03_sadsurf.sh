#!/bin/bash
# Round 3: SAD surfaces on the GPU — parity, then the 1080p encode with and without them
set -u
OUT=gpurun_out/r03_ss
mkdir -p $OUT
timeout 900 python -m pytest tests/test_sadsurf.py tests/test_la_session.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c1080.yuv', 1920, 1080, 120, seed=4321)
PY
R=oracle/_ref
A="--input /tmp/c1080.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
$R/x265_8bit $A -o /tmp/ref.hevc 2>&1 | grep "^encoded" | tee $OUT/ref.log
for tag in sad1 sad0 sad1b; do
    v=1; [ $tag = sad0 ] && v=0
    for i in 1 2; do
        X265HIP_SADPLANES=$v X265HIP=require X265HIP_VERBOSE=1 timeout 300 $R/x265_hip_8bit $A -o /tmp/$tag.hevc 2> $OUT/$tag.$i.log
        grep -h "^encoded\|x265hip: sadplanes\|x265hip: lookahead: .* seam" $OUT/$tag.$i.log
    done
    cmp /tmp/ref.hevc /tmp/$tag.hevc && echo "$tag identical" | tee -a $OUT/identity.txt
done
