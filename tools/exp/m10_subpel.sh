set -u
python - <<PY
import sys; sys.path.insert(0,".")
from x265_amd.synth import make_clip
make_clip("/tmp/m10.yuv",1920,1080,40,seed=4321)
PY
for v in 1 0; do
  mkdir -p gpurun_out/s2_m10_$v
  (cd /tmp && export TMPDIR=/tmp && X265HIP=require X265HIP_SUBPEL_LDS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s2_m10_$v -o p -- $GRAFT_REPO_ROOT/integration/_build/x265_hip_10bit --input /tmp/m10.yuv --input-res 1920x1080 --fps 30 --frames 40 --preset medium --me hex -o /tmp/m10_$v.hevc > /dev/null 2>&1)
  echo "LDS=$v"; grep -E "subpel_satd" $(find gpurun_out/s2_m10_$v -name "*kernel_stats.csv" | head -1) | cut -c1-140
done
cmp /tmp/m10_1.hevc /tmp/m10_0.hevc && echo "same bitstream"
