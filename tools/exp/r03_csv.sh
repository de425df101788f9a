#!/bin/bash
set -u
OUT=gpurun_out/r03_csv
mkdir -p $OUT
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c1080.yuv', 1920, 1080, 120, seed=4321)
PY
R=oracle/_ref
A="--input /tmp/c1080.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
for tag in base sad0 p32; do
  extra=""; env=""
  [ $tag = sad0 ] && export X265HIP_SADPLANES=0
  [ $tag = p32 ] && extra="--pools 32"
  X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A $extra --csv $OUT/$tag.csv --csv-log-level 2 -o /tmp/a.hevc 2>&1 | grep "^encoded\|frame threads\|x265 \[info\]: Thread"
  unset X265HIP_SADPLANES
  python3 - $OUT/$tag.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
hdr = [h.strip() for h in rows[0]]
data = [r for r in rows[1:] if len(r) == len(hdr) and r[0].strip().isdigit()]
def col(name):
    i = hdr.index(name)
    return [float(r[i]) for r in data if r[i].strip() not in ("", "-")]
for t in ("I-SLICE", "P-SLICE", "B-SLICE", "b-SLICE"):
    sel = [r for r in data if r[hdr.index("Type")].strip() == t]
    if not sel: continue
    out = [t, len(sel)]
    for name in ("Total frame time (ms)", "Wall time (ms)", "Ref Wait Wall (ms)", "Total CTU time (ms)", "Stall Time (ms)", "Avg WPP", "Row Blocks", "DecideWait (ms)", "Row0Wait (ms)"):
        if name in hdr:
            i = hdr.index(name)
            v = [float(r[i]) for r in sel]
            out.append("%s %.1f" % (name.split(" (")[0], sum(v) / len(v)))
    print(" | ".join(map(str, out)))
PY
done
