#!/bin/bash
set -u
python3 - <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import fuzz_encoder as fz
from x265_amd.synth import make_clip
c = fz.draw(59)
make_clip('/tmp/f59.yuv', c["width"], c["height"], c["frames"], seed=1000 + 59, tile=48, vmax=7, fade=c["fade"], csp=c["csp"], depth=8)
open('/tmp/f59.args', 'w').write(" ".join(["--input", "/tmp/f59.yuv", "--input-res", "%dx%d" % (c["width"], c["height"]), "--input-depth", "8", "--input-csp", c["csp"], "--fps", "30", "--frames", str(c["frames"]), "--hash", "1"] + c["args"]))
PY
R=oracle/_ref
A=$(cat /tmp/f59.args)
$R/x265_8bit $A -o /tmp/ref.hevc > /dev/null 2>&1
X265HIP_VERIFY=1 X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/b.hevc > /tmp/b.log 2>&1; echo "rc $?"
grep -v "^x265 \[info\]\|^yuv\|^raw" /tmp/b.log | tail -12 | cut -c1-300
echo "--- without verify, weighted mirrors only suspects: X265HIP_SADPLANES_LEVELS=12 (no 16x16 lookups)"
X265HIP_SADPLANES_LEVELS=12 X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/b.hevc > /tmp/b.log 2>&1; echo "rc $?"; cmp -s /tmp/ref.hevc /tmp/b.hevc && echo identical || echo DIFFERENT
grep "sadplanes\|refplanes" /tmp/b.log | cut -c1-300
