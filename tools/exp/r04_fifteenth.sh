#!/bin/bash
# round 4, fifteenth GPU call: stage stamps of the transform chain; the server's poll without the per-poll PCIe read of `leave`
set -u
OUT=gpurun_out/r04_o
mkdir -p $OUT
timeout 300 python -m pytest tests/test_cuserve.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/pytest.txt
timeout 120 tools/micro/cuserve_rt 0 2000 1 2>&1 | cut -c1-400 | tee $OUT/cuserve_rt_stamps.txt
timeout 120 tools/micro/cuserve_rt 0 2000 0 2>&1 | cut -c1-300 | tee $OUT/cuserve_rt.txt
