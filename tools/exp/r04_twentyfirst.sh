#!/bin/bash
# round 4, twenty-first GPU call: one launch for surfaces of several reference pictures (per-job reference pointer, attach jobs batched by the worker)
set -u
OUT=gpurun_out/r04_u
mkdir -p $OUT
timeout 600 python -m pytest tests/test_sadsurf.py tests/test_places.py tests/test_refpic.py tests/test_x265_dropin.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 200 $OUT/bench_line.json
