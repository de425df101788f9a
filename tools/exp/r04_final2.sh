#!/bin/bash
# round 4, after the Main10 sse_ss fix: the whole GPU test tier and smoke again; CU jobs one launch each under rocprofv3
set -u
OUT=gpurun_out/r04_z
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/gpu_test_tier.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out; tag=r04_v1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_encode_mode1 -o e -- $root/tools/micro/cuserve_rt 1 400 > $out/${tag}_encode_mode1.log 2>&1
find $out/${tag}_encode_mode1 -name "*kernel_trace.csv" -delete
tail -3 $out/${tag}_encode_mode1.log | cut -c1-200
