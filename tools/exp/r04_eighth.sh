#!/bin/bash
# round 4, eighth GPU call: the whole GPU test tier after the sub-pel kernel's out-of-picture fix
set -u
OUT=gpurun_out/r04_h
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $OUT/gpu_test_tier_full.txt 2>&1
tail -15 $OUT/gpu_test_tier_full.txt
