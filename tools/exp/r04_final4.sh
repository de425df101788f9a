#!/bin/bash
# round 4, closing call on the final tree: GPU test tier, smoke, rocprofv3 evidence (stamped with the final kernel sources), bench lines, A/B
set -u
OUT=gpurun_out/r04_z
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/gpu_test_tier.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1500 bash tools/collect_profiles_r04.sh r04_v1 > $OUT/collect.log 2>&1
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 300 $OUT/bench_line.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_args.json 2> $OUT/bench_err2.txt; tail -c 300 $OUT/bench_line_driver_args.json
R3="X265HIP_CUSERVE=0,X265HIP_SADPLANES_SUBPEL=0,X265HIP_SADPLANES_RECT=0,X265HIP_SADSURF_BATCH=0,X265HIP_SADSURF_ROUNDS=0,X265HIP_SADSURF_GATHER_US=0,X265HIP_PINNED=hip"
timeout 900 python tools/ab_encode.py --rounds 3 --frames 120 on: r3:$R3 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
du -sh gpurun_out
