python - <<'PY'
import os, sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c2.yuv', 3840, 2160, 8, seed=31)
PY
A="--input /tmp/c2.yuv --input-res 3840x2160 --fps 30 --frames 8 --preset slow --me star --merange 57 --hash 1"
oracle/_ref/x265_8bit $A -o /tmp/ref.hevc > /dev/null 2>&1
for cfg in "X=1" "X=1" "X=1" "X265HIP_CUSERVE_TEAM=0" "X265HIP_CUSERVE_TEAM=0" "X265HIP_INTRASCAN=0" "X265HIP_INTRASCAN=0" "X265HIP_SAOSTATS=0" "X265HIP_SAOSTATS=0" "X265HIP_CUSERVE_SLOTS=1" ; do
  for v in 1 0; do
    env $cfg X265HIP=require $( [ $v = 1 ] && echo X265HIP_VERIFY=1 ) integration/_build/x265_hip_8bit $A -o /tmp/g.hevc > /tmp/g.log 2>&1; rc=$?
    echo "$cfg verify=$v rc=$rc same=$(cmp -s /tmp/ref.hevc /tmp/g.hevc && echo yes || echo NO) $(grep -h "VERIFY\|differ" /tmp/g.log | head -2 | cut -c1-200)"
  done
done
