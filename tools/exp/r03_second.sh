#!/bin/bash
# Round 3: the register-blocked search-window kernel — parity, the kernel on its own (tools/sadsurf_bench.py), instruction rates, A/B of the encode
set -u
OUT=gpurun_out/r03_b
mkdir -p $OUT
export TMPDIR=/tmp
tools/micro/qsad_rate 2>&1 | tee $OUT/qsad_rate.txt
timeout 600 python -m pytest tests/test_sadsurf.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
timeout 300 python tools/sadsurf_bench.py 2>&1 | tee $OUT/sadsurf_bench.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/ss_stats -o s -- python $OLDPWD/tools/sadsurf_bench.py --modes batch > $OLDPWD/$OUT/ss_stats.log 2>&1)
find $OUT/ss_stats -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 base: sad0:X265HIP_SADPLANES=0 --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
for tag in base p32; do
  extra=""; [ $tag = p32 ] && extra="--pools 32"
  X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A $extra --csv $OUT/$tag.csv --csv-log-level 2 -o /tmp/a.hevc 2>&1 | grep "^encoded\|frame threads\|x265 \[info\]: Thread" | tee -a $OUT/csv_summary.txt
  python3 - $OUT/$tag.csv <<'PY' | tee -a $OUT/csv_summary.txt
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
hdr = [h.strip() for h in rows[0]]
data = [r for r in rows[1:] if len(r) == len(hdr) and r[0].strip().isdigit()]
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
