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
X265HIP_VERIFY=1 X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/b.hevc 2>&1 | grep "x265hip\|VERIFY\|encoded" | cut -c1-260
cmp /tmp/ref.hevc /tmp/b.hevc && echo identical
for extra in "--no-weightp" "--ctu 32" "--ctu 64"; do
  $R/x265_8bit $A $extra -o /tmp/ref2.hevc > /dev/null 2>&1
  X265HIP_VERIFY=1 X265HIP=require timeout 120 $R/x265_hip_8bit $A $extra -o /tmp/b2.hevc 2>&1 | grep "VERIFY" | cut -c1-260
  cmp -s /tmp/ref2.hevc /tmp/b2.hevc && echo "$extra identical" || echo "$extra DIFFERENT"
done
