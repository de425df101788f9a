#!/bin/bash
# Round-3 rocprofv3 evidence for bench.py's roofline blocks (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles_r03.sh r03_v3'
# Kernel trace + stats in one run; FETCH_SIZE, WRITE_SIZE and the SQ counters each in their own --pmc run (never combined with other trace
# domains).  Raw output goes to gpurun_out/<tag>_*; `python tools/summarise_profiles_r03.py <tag>` turns it into profiles/<tag>_* (stamped with the
# sha256 of the kernel sources, which bench.py compares with its own tree before it quotes a profile).
#   *_ss*     the search-window kernel on whole pictures: python tools/sadsurf_bench.py --modes frame
#   *_la*     lookahead_p_kernel at the encode's launch size: python bench.py --lookahead-probe-only --probe-pairs 35
#   *_encode  the real encode (integration/_build/x265_hip_8bit, 120 frames): which kernels the encoder's GPU work consists of
#   calib_*   known-byte-count kernels (tools/pmc_calibrate.py) for the byte counters' correction factors
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
prof() { # name, then the command; passes: stats, fetch, write, sq
    name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}stats -o p -- "$@" > $out/${tag}_${name}stats.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_${name}fetch -o p -- "$@" > $out/${tag}_${name}fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_${name}write -o p -- "$@" > $out/${tag}_${name}write.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
        --kernel-trace --output-format csv -d $out/${tag}_${name}sq -o p -- "$@" > $out/${tag}_${name}sq.log 2>&1
}
prof ss python $root/tools/sadsurf_bench.py --modes frame --reps 3
prof la python $root/bench.py --lookahead-probe-only --probe-pairs 35
rm -rf $out/calib_fetch $out/calib_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/calib_fetch -o c -- python $root/tools/pmc_calibrate.py > $out/${tag}_calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/calib_write -o c -- python $root/tools/pmc_calibrate.py > $out/${tag}_calib_write.log 2>&1
python -c "
import sys; sys.path.insert(0, '$root')
from x265_amd.synth import make_clip
make_clip('/tmp/prof_clip.yuv', 1920, 1080, 120, seed=4321)"
X265HIP=require X265HIP_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_encode -o e -- $root/integration/_build/x265_hip_8bit --input /tmp/prof_clip.yuv \
    --input-res 1920x1080 --fps 30 --preset medium --me hex --frames 120 -o /dev/null > $out/${tag}_encode.log 2>&1
find $out/${tag}_encode -name "*kernel_trace.csv" -size +30M -delete
cd $root
ls -d $out/${tag}_* | head -40
