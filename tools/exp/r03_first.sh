#!/bin/bash
# Round 3, first measurement of the tree after the re-entry: new parity tests, interleaved A/B of the seams, kernel stats of the encode,
# SAD-lookup cycle timing, lookahead timeline, bench line
set -u
OUT=gpurun_out/r03_a
mkdir -p $OUT
export TMPDIR=/tmp
git rev-parse HEAD > $OUT/commit.txt 2>/dev/null
timeout 600 python -m pytest tests/test_sadsurf.py tests/test_la_session.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
timeout 600 python tools/ab_encode.py --rounds 3 --frames 120 base: sad0:X265HIP_SADPLANES=0 ahead0:X265HIP_LOOKAHEAD_AHEAD=0 \
    both0:X265HIP_SADPLANES=0,X265HIP_LOOKAHEAD_AHEAD=0 --out $OUT/ab.json 2>&1 | tee $OUT/ab.txt
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
for v in 1 2; do
  X265HIP_DEBUG_SADTIME=$v X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/t.hevc 2>&1 | grep "^encoded\|x265hip: sadplanes" | tee -a $OUT/sadtime.txt
done
X265HIP_DEBUG_LA_TIMELINE=/tmp/tl.txt X265HIP=require X265HIP_VERBOSE=1 timeout 120 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded\|x265hip:" | tee $OUT/timeline_run.txt
python3 tools/la_timeline.py /tmp/tl.txt 2>&1 | tee $OUT/timeline.txt
(cd /tmp && X265HIP=require X265HIP_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/../../$OUT/enc_stats -o e -- $R/x265_hip_8bit $A -o /dev/null > $R/../../$OUT/enc_stats.log 2>&1)
grep "^encoded\|x265hip:" $OUT/enc_stats.log
find $OUT/enc_stats -name "*kernel_stats.csv" | head -1 | xargs cat | head -30
find $OUT/enc_stats -name "*kernel_trace.csv" -size +20M -delete
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
