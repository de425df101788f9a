#!/bin/bash
# bisect the GPU-tier mismatch of fuzz seed 59 (200x136, --ctu 16 --tune animation --aq-mode 3 -F 1 --pools 6): which seam?
set -u
OUT=gpurun_out/r03_seed59
mkdir -p $OUT
python3 - <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import fuzz_encoder as fz
from x265_amd.synth import make_clip
c = fz.draw(59)
print(c)
make_clip('/tmp/f59.yuv', c["width"], c["height"], c["frames"], seed=1000 + 59, tile=48, vmax=7, fade=c["fade"], csp=c["csp"], depth=8)
open('/tmp/f59.args', 'w').write(" ".join(["--input", "/tmp/f59.yuv", "--input-res", "%dx%d" % (c["width"], c["height"]), "--input-depth", "8", "--input-csp", c["csp"], "--fps", "30", "--frames", str(c["frames"]), "--hash", "1"] + c["args"]))
PY
R=oracle/_ref
A=$(cat /tmp/f59.args)
$R/x265_8bit $A -o /tmp/ref.hevc > /dev/null 2>&1
run() { tag=$1; shift
  for i in 1 2 3; do
    env "$@" X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/b.hevc > /tmp/b.log 2>&1
    if cmp -s /tmp/ref.hevc /tmp/b.hevc; then echo "$tag run $i: identical"; else echo "$tag run $i: DIFFERENT rc=$?"; grep "x265hip:" /tmp/b.log | cut -c1-200 | head -8; fi
  done
}
run all            X265HIP_NOP=1
run no_sadplanes   X265HIP_SADPLANES=0
run no_refplanes   X265HIP_REFPLANES=0
run no_srcplanes   X265HIP_SRCPLANES=0
run no_lookahead   X265HIP_LOOKAHEAD=0
run verify         X265HIP_VERIFY=1
run no_ahead       X265HIP_LOOKAHEAD_AHEAD=0
