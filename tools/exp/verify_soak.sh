#!/bin/bash
# tools/exp/verify_soak.sh <tag> — X265HIP_VERIFY=1 (every value a seam serves is recomputed by the reference's own function and compared; a difference aborts) over longer encodes than the
# GPU test tier's, each bitstream compared with the unmodified reference's: 1080p medium 240 frames, 4K slow 16 frames, 4K Main10 slower 8 frames, 8K medium 8 frames
set -u
OUT=gpurun_out/$1; mkdir -p $OUT
clip() { python - "$1" "$2" "$3" "$4" <<'PY'
import os, sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
if not os.path.exists(sys.argv[1]): make_clip(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), seed=4321)
PY
}
run() { # name bits res frames args...
  name=$1; bits=$2; res=$3; frames=$4; shift 4
  w=${res%x*}; h=${res#*x}
  clip /tmp/vs_$name.yuv $w $h $frames
  a=(--input /tmp/vs_$name.yuv --input-res $res --fps 30 --frames $frames --hash 1 "$@")
  oracle/_ref/x265_${bits}bit "${a[@]}" -o /tmp/vs_ref.hevc > /dev/null 2>&1
  t0=$(date +%s)
  X265HIP=require X265HIP_VERIFY=1 X265HIP_VERBOSE=1 integration/_build/x265_hip_${bits}bit "${a[@]}" -o /tmp/vs_hip.hevc > $OUT/$name.log 2>&1; rc=$?
  same=no; cmp -s /tmp/vs_ref.hevc /tmp/vs_hip.hevc && same=yes
  echo "$name: $res $frames frames ${bits}-bit $*: exit $rc, bitstream identical to the reference's: $same, $(( $(date +%s) - t0 )) s under VERIFY; $(grep -c 'VERIFY FAILED' $OUT/$name.log) VERIFY failures; $(grep -E 'cuserve: [0-9]+ CU' $OUT/$name.log | cut -c1-120)" | tee -a $OUT/verify_soak.txt
}
run 1080p_medium 8 1920x1080 240 --preset medium --me hex --pools 24 -F 6
run 4k_slow 8 3840x2160 16 --preset slow --me star --merange 57 --pools 24 -F 6
run 4k_main10_slower 10 3840x2160 8 --preset slower --rd 6 --pools 24 -F 6
run 8k_medium 8 7680x4320 8 --preset medium --me hex --pools 24 -F 6
