#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r01_v4'
# Kernel trace + stats in one run; FETCH_SIZE, WRITE_SIZE and the SQ counters each in their own --pmc run (never combined with
# other trace domains).  Raw output goes to gpurun_out/<tag>_*; tools/summarise_profiles.py turns it into profiles/<tag>_*.
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="python $root/bench.py --steps 10 --warmup 2 --cpu-frames 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o b -- $cmd > $out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -o b -- $cmd > $out/${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -o b -- $cmd > $out/${tag}_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $out/${tag}_sq -o b -- $cmd > $out/${tag}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_lookahead -o b -- env CPU=0 PAIRS=64 python $root/tools/lookahead_bench.py > $out/${tag}_lookahead.log 2>&1
cd $root
python bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench_line.json
ls $out/${tag}_*
