#!/bin/bash
# round 4, closing GPU call on the final tree: GPU test tier, smoke, rocprofv3 evidence, bench lines, hand-off measurements, A/B, CPU profile
set -u
OUT=gpurun_out/r04_z
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/gpu_test_tier.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1500 bash tools/collect_profiles_r04.sh ${1:-r04_v1} > $OUT/collect.log 2>&1
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 300 $OUT/bench_line.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_args.json 2> $OUT/bench_err2.txt; tail -c 300 $OUT/bench_line_driver_args.json
for m in device host; do X265HIP_CUSERVE_MAILBOX=$m timeout 120 tools/micro/cuserve_rt 0 3000 2>&1 | cut -c1-300 > $OUT/cuserve_rt_mailbox_$m.txt; done
timeout 120 tools/micro/cuserve_rt 0 2000 1 2>&1 | cut -c1-400 > $OUT/cuserve_rt_stamps.txt
timeout 120 tools/micro/cuserve_rt 1 2000 2>&1 | cut -c1-300 > $OUT/cuserve_rt_mode1.txt
timeout 60 tools/micro/create_cost > $OUT/create_cost.txt 2>&1
R3="X265HIP_CUSERVE=0,X265HIP_SADPLANES_SUBPEL=0,X265HIP_SADPLANES_RECT=0,X265HIP_SADSURF_BATCH=0,X265HIP_SADSURF_ROUNDS=0,X265HIP_PINNED=hip"
timeout 900 python tools/ab_encode.py --rounds 4 --frames 120 on: r3:$R3 --out $OUT/ab1080.json 2>&1 | tee $OUT/ab1080.txt
R=$(pwd)/oracle/_ref
A="--input /tmp/ab_clip_1920x1080_120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
X265HIP_CPUSAMPLE_OUT=/tmp/hip.bin LD_PRELOAD=$(pwd)/tools/prof/libcpusample.so X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2>&1 | grep "^encoded"
python3 tools/prof/resolve.py /tmp/hip.bin 120 > $OUT/cpu_profile_bound_encoder.txt
X265HIP=require X265HIP_VERBOSE=1 X265HIP_DEBUG_STARTUP=1 $R/x265_hip_8bit $A -o /dev/null 2>&1 | grep "startup\|^encoded" | grep -v "PicYuv::destroy\|device copy of a source" > $OUT/startup_marks.txt
du -sh gpurun_out
