#!/bin/bash
# round 4, eighteenth GPU call: SAD-surface launches in whole rounds from every path (attach included): tests + the live VALU fraction in the bench line
set -u
OUT=gpurun_out/r04_r
mkdir -p $OUT
timeout 600 python -m pytest tests/test_sadsurf.py tests/test_places.py tests/test_refpic.py tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 300 $OUT/bench_line.json
