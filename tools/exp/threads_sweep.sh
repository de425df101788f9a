#!/bin/bash
# tools/exp/threads_sweep.sh <out> — bound encoder and reference over (--pools, --frame-threads) on the bench clip (120 frames): which thread arguments each
# encoder is fastest with on a box whose cgroup quota (16 CPUs) is far below the cores it shows (256).  CELLS="pools:F ..." (default: a 5 x 4 grid), RUNS (2);
# the runs of all cells are interleaved; mean and best fps per cell; the two encoders' bitstreams of a cell must agree (md5).
OUT=$1; mkdir -p $(dirname $OUT)
CELLS=${CELLS:-"12:4 16:4 20:4 24:4 32:4 12:5 16:5 20:5 24:5 32:5 12:6 16:6 20:6 24:6 32:6 12:8 16:8 20:8 24:8 32:8"}
RUNS=${RUNS:-2}
python - <<'PY'
import os, sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
if not os.path.exists('/tmp/sweep120.yuv'): make_clip('/tmp/sweep120.yuv', 1920, 1080, 120, seed=4321)
PY
D=/tmp/sweep_$$; mkdir -p $D
for k in $(seq 1 $RUNS); do for cell in $CELLS; do P=${cell%%:*}; F=${cell##*:}
  for b in hip ref; do
    exe=integration/_build/x265_hip_8bit; [ $b = ref ] && exe=oracle/_ref/x265_8bit
    f=$(X265HIP=require $exe --input /tmp/sweep120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --pools $P --frame-threads $F -o $D/$b.hevc 2>&1 | grep -E "^encoded" | sed -E 's/.*\(([0-9.]+) fps\).*/\1/')
    echo "${f:-0} $(md5sum < $D/$b.hevc | cut -c1-8)" >> $D/${b}_${P}_${F}.txt
  done
done; done
{
echo "# pools frame-threads : bound fps mean / best | reference fps mean / best | same bitstream   ($RUNS interleaved runs, 120 frames 1080p medium hex)"
for cell in $CELLS; do P=${cell%%:*}; F=${cell##*:}
  python - $D $P $F <<'PY'
import sys
d, p, f = sys.argv[1:]
def rd(b):
    rows = [l.split() for l in open("%s/%s_%s_%s.txt" % (d, b, p, f))]
    v = [float(r[0]) for r in rows]
    return sum(v) / len(v), max(v), {r[1] for r in rows}
h, r = rd("hip"), rd("ref")
print("%3s %2s : %6.2f / %6.2f | %6.2f / %6.2f | %s" % (p, f, h[0], h[1], r[0], r[1], h[2] == r[2] and len(h[2]) == 1))
PY
done
} | tee $OUT
rm -rf $D
