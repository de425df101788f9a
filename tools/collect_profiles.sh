#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02_v1'
# Kernel trace + stats in one run; FETCH_SIZE, WRITE_SIZE and the SQ counters each in their own --pmc run (never combined with
# other trace domains).  Raw output goes to gpurun_out/<tag>_*; tools/summarise_profiles.py turns it into profiles/<tag>_*.
#   *_stats / _fetch / _write / _sq   the device-resident frame pass (bench.py's `frame_pass` block): python tools/framepass_profile_cmd.py
#   *_lookahead                        the lookahead cost pass, 64 pairs per launch
#   *_lastats / _lafetch / _lawrite / _lasq   bench.py's own lookahead_p_kernel probe (8 pairs per launch): python bench.py --lookahead-probe-only
#   *_encode                           the real encode (integration/_build/x265_hip_8bit, 60 frames): which kernels the encoder's GPU work consists of
#   calib_fetch / calib_write          known-byte-count kernels (tools/pmc_calibrate.py) for the byte counters' correction factors
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="python $root/bench.py --steps 1 --warmup 0 --no-ref-encoder --cpu-frames 0 --frame-pass-only"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o b -- $cmd > $out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -o b -- $cmd > $out/${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -o b -- $cmd > $out/${tag}_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $out/${tag}_sq -o b -- $cmd > $out/${tag}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_lookahead -o b -- env CPU=0 PAIRS=64 python $root/tools/lookahead_bench.py > $out/${tag}_lookahead.log 2>&1
# the lookahead_p_kernel probe of bench.py's roofline block (8 pairs per launch), byte counters and SQ counters in their own passes
lacmd="python $root/bench.py --lookahead-probe-only"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_lastats -o b -- $lacmd > $out/${tag}_lastats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_lafetch -o b -- $lacmd > $out/${tag}_lafetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_lawrite -o b -- $lacmd > $out/${tag}_lawrite.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $out/${tag}_lasq -o b -- $lacmd > $out/${tag}_lasq.log 2>&1
rm -rf $out/calib_fetch $out/calib_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/calib_fetch -o c -- python $root/tools/pmc_calibrate.py > $out/${tag}_calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/calib_write -o c -- python $root/tools/pmc_calibrate.py > $out/${tag}_calib_write.log 2>&1
python -c "
import sys; sys.path.insert(0, '$root')
from x265_amd.synth import make_clip
make_clip('/tmp/prof_clip.yuv', 1920, 1080, 60, seed=4321)"
X265HIP_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_encode -o e -- $root/integration/_build/x265_hip_8bit --input /tmp/prof_clip.yuv \
    --input-res 1920x1080 --fps 30 --preset medium --me hex --frames 60 -o /dev/null > $out/${tag}_encode.log 2>&1
cd $root
ls $out/${tag}_* | head -40
