#!/bin/bash
# tools/exp/gpu.sh — ONE parameterised script for what is run on the MI355X box through gpurun (replaces the per-experiment r03_* / r04_* scripts; those
# are in the history).  Everything lands under gpurun_out/<tag>/; summaries worth keeping are copied into profiles/.
#
#   tools/exp/gpu.sh <tag> <task> [task ...]
#     tests [-k expr]      the GPU test tier (or a selection)
#     bench                python bench.py (the contract line)
#     configs23            BASELINE configs[2] / configs[3] at 4K: defaults (coefficient-mode CU jobs) vs X265HIP_CUSERVE_RDOQ=0 (RDOQ CUs stay on the host)
#     ab <frames> <name:ENV=..,..> ...   interleaved A/B of the 1080p preset-medium bench encode
#     rt                   tools/micro/cuserve_rt: the job round trip, with stage stamps
#     stats                rocprofv3 --kernel-trace --stats of a 120-frame bound encode
#     pmc                  FETCH_SIZE / WRITE_SIZE per job: cuserve_rt in launch mode (one dispatch = one job), one shape per pass, counters in their own passes
#     stress <n>           two_encoders_hip8 par looped n times (encoders alive at the same time on the real library; every other run with poisoned Analysis
#                          objects, every third with X265HIP_REFPLANES=0), each session's bitstream compared with the reference objects' (tests/test_reference_races.py)
#     framestats           x265's own per-frame clocks (--csv-log-level 2: DecideWait, Row0Wait, Wall, Ref Wait Wall, Total CTU time, Stall, Avg WPP, Row Blocks)
#                          of the bound encoder and of the reference on the bench clip: where a frame encoder's wall clock goes
#     callers <regex>      who calls the functions matching <regex>: one 240-frame run under the sampler with call chains (X265HIP_CPUSAMPLE_STACK=1), tools/prof/callers.py
#     pmc_encode           FETCH_SIZE / WRITE_SIZE per launch of the kernels launched in a 30-frame bound encode (job server off), one pass per counter
#     cpuprofile           the bound encoder under the CPU sampler (tools/prof), 6 x 240 frames merged (> 10 k samples)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export X265HIP_VERBOSE=1
clip() { python - "$1" "$2" <<'PY'
import os, sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
if not os.path.exists(sys.argv[1]): make_clip(sys.argv[1], 1920, 1080, int(sys.argv[2]), seed=4321)
PY
}
while [ $# -gt 0 ]; do
  task=$1; shift
  case $task in
    tests)
      sel=(); if [ "${1:-}" = "-k" ]; then sel=(-k "$2"); shift 2; fi
      timeout 1500 python -m pytest tests -m gpu -x -q "${sel[@]}" 2>&1 | tail -15 > $OUT/gputest.log; tail -3 $OUT/gputest.log ;;
    bench)
      timeout 600 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json ;;
    configs23)
      # BASELINE configs[2] / configs[3] as JSON lines shaped like bench.py's (value, cpu_baseline with build flags, byte identity, served counts): CFG_ROUNDS
      # interleaved rounds (default 5), the bench line's thread arguments; then the coefficient-mode A/B (RDOQ CUs on the host vs as jobs)
      TH="--pools 24 --frame-threads 6"
      timeout 1500 python tools/config_bench.py --encode configs2 --rounds ${CFG_ROUNDS:-5} --threads "$TH" 2> $OUT/configs2.err | tail -1 > $OUT/configs2_line.json; cut -c1-400 $OUT/configs2_line.json
      timeout 1500 python tools/config_bench.py --encode configs3 --rounds ${CFG_ROUNDS:-5} --threads "$TH" 2> $OUT/configs3.err | tail -1 > $OUT/configs3_line.json; cut -c1-400 $OUT/configs3_line.json
      if [ "${CFG_AB:-1}" = 1 ]; then
        OFF="X265HIP_CUSERVE_RDOQ=0"; NOINV="X265HIP_CUSERVE_INVERSE=0"
        timeout 2400 python tools/ab_encode.py --rounds ${CFG_ROUNDS:-5} --frames 24 --res 3840x2160 --preset slow --extra "--me star --merange 57 $TH" base: noinv:$NOINV nocoef:$OFF --out $OUT/configs2.json 2>&1 | tee $OUT/configs2_4k_slow_star_ab.txt | tail -6
        timeout 2400 python tools/ab_encode.py --rounds ${CFG_ROUNDS:-5} --frames 8 --res 3840x2160 --preset slower --extra "--rd 6 $TH" --bits 10 base: noinv:$NOINV nocoef:$OFF --out $OUT/configs3.json 2>&1 | tee $OUT/configs3_4k_main10_slower_ab.txt | tail -6
      fi ;;
    ab)
      frames=$1; shift; cfgs=(); while [ $# -gt 0 ] && [[ "$1" == *:* ]]; do cfgs+=("$1"); shift; done
      # AB_ROUNDS (default 3), AB_EXTRA (default "--me hex"; the bench line's arguments on the MI355X box are "--me hex --pools 16 --frame-threads 5"), AB_NAME (file stem)
      timeout ${AB_TIMEOUT:-1500} python tools/ab_encode.py --rounds ${AB_ROUNDS:-3} --frames $frames --extra "${AB_EXTRA:---me hex}" "${cfgs[@]}" --out $OUT/${AB_NAME:-ab}.json 2>&1 | tee $OUT/${AB_NAME:-ab}.txt | tail -${AB_TAIL:-14} ;;
    rt)
      (timeout 200 tools/micro/cuserve_rt 0 3000 1; timeout 100 tools/micro/cuserve_rt 1 1000 0) 2>&1 | tee $OUT/cuserve_rt.txt | tail -30 ;;
    stats)
      clip /tmp/bench120.yuv 120
      HERE=$PWD
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $HERE/$OUT/prof -o p -- $HERE/integration/_build/x265_hip_8bit --input /tmp/bench120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex -o /tmp/s.hevc > $HERE/$OUT/stats_run.log 2>&1)
      find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-180 ;;
    stress)
      n=$1; shift
      R=oracle/_ref; D=/tmp/stress_$TAG; mkdir -p $D
      X265HIP=require $R/two_encoders_ref8 $D/ref || { echo "reference run failed"; exit 1; }
      read elem off < <($R/tld_layout8)
      bad=0; t0=$(date +%s)
      for i in $(seq 1 $n); do
        envs=(X265HIP=require)
        [ $((i % 2)) = 1 ] && envs+=(LD_PRELOAD=$R/allocshim.so ALLOCSHIM_SIZE=$((8 + 4 * elem)) ALLOCSHIM_ELEM=$elem ALLOCSHIM_OFFSET=$off ALLOCSHIM_WORD=2)
        [ $((i % 3)) = 0 ] && envs+=(X265HIP_REFPLANES=0)
        env "${envs[@]}" TWO_ENCODERS_WATCHDOG=120 timeout 200 integration/_build/two_encoders_hip8 $D/g par > $D/run.log 2>&1 || { echo "run $i: exit $?"; tail -5 $D/run.log; bad=$((bad + 1)); continue; }
        for k in 0 1 2 3 4; do cmp -s $D/ref_$k.hevc $D/g_$k.hevc || { echo "run $i: session $k differs"; bad=$((bad + 1)); }; done
      done
      echo "concurrent encoders on the MI355X: $n runs (5 sessions each, two then three alive at a time), $bad mismatches or failures, $(( $(date +%s) - t0 )) s" | tee $OUT/concurrent_stress.txt ;;
    framestats)
      clip /tmp/bench120.yuv 120
      for b in hip ref; do
        exe=integration/_build/x265_hip_8bit; [ $b = ref ] && exe=oracle/_ref/x265_8bit
        X265HIP=require $exe --input /tmp/bench120.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --csv $OUT/framestats_$b.csv --csv-log-level 2 -o /tmp/f.hevc 2>&1 | grep -E "^encoded" | tee $OUT/framestats_$b.log
      done
      python - $OUT <<'PY' | tee $OUT/frame_clocks.txt
import csv, sys
for b in ("hip", "ref"):
    rows = [r for r in csv.reader(open("%s/framestats_%s.csv" % (sys.argv[1], b)))]
    head = [h.strip() for h in rows[0]]
    cols = ["DecideWait (ms)", "Row0Wait (ms)", "Wall time (ms)", "Ref Wait Wall (ms)", "Total CTU time (ms)", "Stall Time (ms)", "Total frame time (ms)", "Avg WPP", "Row Blocks"]
    body = [r for r in rows[1:] if len(r) >= len(head) and r[0].strip().isdigit()]
    print(b, len(body), "frames; per-frame means:")
    for c in cols:
        if c in head:
            i = head.index(c)
            v = [float(r[i]) for r in body if r[i].strip()]
            print("   %-24s %9.2f" % (c, sum(v) / max(1, len(v))))
    for t in ("I-SLICE", "P-SLICE", "B-SLICE", "b-SLICE"):
        i, j, k = head.index("Wall time (ms)"), head.index("Total CTU time (ms)"), head.index("Ref Wait Wall (ms)")
        sel = [r for r in body if r[1].strip() == t]
        if sel:
            print("   %s x %d: wall %.1f ms, CTU time %.1f ms, ref wait wall %.1f ms" % (t, len(sel), sum(float(r[i]) for r in sel) / len(sel), sum(float(r[j]) for r in sel) / len(sel), sum(float(r[k]) for r in sel) / len(sel)))
PY
      ;;
    callers)
      pat=$1; shift
      clip /tmp/bench240.yuv 240
      X265HIP_CPUSAMPLE_STACK=1 X265HIP_CPUSAMPLE_OUT=/tmp/cs.bin LD_PRELOAD=tools/prof/libcpusample.so integration/_build/x265_hip_8bit --input /tmp/bench240.yuv --input-res 1920x1080 --fps 30 --frames 240 --preset medium --me hex --pools 24 -F 6 -o /tmp/p.hevc 2>&1 | grep -E "^encoded" > $OUT/callers_run.log
      python tools/prof/callers.py /tmp/cs.bin "$pat" 25 > $OUT/callers.txt 2>&1; head -60 $OUT/callers.txt | cut -c1-250 ;;
    pmc)
      HERE=$PWD
      for shape in cu5 cu6 sao; do for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
        (cd /tmp && export TMPDIR=/tmp && CUSERVE_RT_ONLY=$shape CUSERVE_RT_THREADS=1 timeout 200 rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $HERE/$OUT/pmc/${shape}_${c%%:*} -o p -- $HERE/tools/micro/cuserve_rt 1 400 0 > $HERE/$OUT/pmc_${shape}_${c%%:*}.log 2>&1)
      done; done
      # the resident server (one dispatch for the whole run, idle polling included): 3100 jobs of one shape from one thread
      for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
        (cd /tmp && export TMPDIR=/tmp && CUSERVE_RT_ONLY=cu5 CUSERVE_RT_THREADS=1 timeout 200 rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $HERE/$OUT/pmc/srv5_${c%%:*} -o p -- $HERE/tools/micro/cuserve_rt 0 3000 0 > $HERE/$OUT/pmc_srv5_${c%%:*}.log 2>&1)
      done
      # algorithmic bytes: pixels in; levels + residual + unit records out (CU jobs), 480 statistics words out (SAO)
      python tools/prof/pmc_launches.py $OUT/pmc cu5:3200:6240 cu6:12416:24960 sao:12675:1920 srv5:3200:6240:3100 | tee $OUT/cuserve_pmc_per_job.txt ;;
    pmc_encode)
      # HBM bytes per launch of the kernels that are LAUNCHED in the bound encode (SAD surfaces, sub-pel SATD tables, phase planes, energy planes, lookahead): two
      # counter passes over a 30-frame encode with the resident job server off (a kernel that stays on the chip would hold up the serialised dispatches of a --pmc run)
      clip /tmp/bench30.yuv 30
      HERE=$PWD
      for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
        (cd /tmp && export TMPDIR=/tmp && X265HIP=require X265HIP_CUSERVE=0 X265HIP_SAOSTATS=0 X265HIP_INTRASCAN=0 timeout 400 rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $HERE/$OUT/pmc_encode/${c%%:*} -o p -- \
           $HERE/integration/_build/x265_hip_8bit --input /tmp/bench30.yuv --input-res 1920x1080 --fps 30 --frames 30 --preset medium --me hex --pools 24 -F 6 -o /tmp/pe.hevc > $HERE/$OUT/pmc_encode_${c%%:*}.log 2>&1)
      done
      find $OUT/pmc_encode -name "*kernel_trace.csv" -delete
      python tools/prof/pmc_kernels.py $OUT/pmc_encode "x265_hip_8bit, 30 frames 1080p preset medium --me hex, CU / SAO / intra jobs off" sadsurf.hip | tee $OUT/pmc_encode.txt | head -16 ;;
    startup)
      # where the wall clock outside x265's own fps clock goes: X265HIP_DEBUG_STARTUP marks (ms since the process started), the process's wall clock as
      # the shell sees it, the encoder's own line — bound encoder and reference, three runs each
      clip /tmp/bench240.yuv 240
      for k in 1 2 3; do for b in hip ref; do
        exe=integration/_build/x265_hip_8bit; [ $b = ref ] && exe=oracle/_ref/x265_8bit
        t0=$(date +%s%N)
        X265HIP=require X265HIP_DEBUG_STARTUP=1 $exe --input /tmp/bench240.yuv --input-res 1920x1080 --fps 30 --frames 240 --preset medium --me hex --pools 24 -F 6 -o /tmp/su.hevc > $OUT/startup_${b}_$k.log 2>&1
        t1=$(date +%s%N)
        echo "$b run $k: wall $(( (t1 - t0) / 1000000 )) ms; $(grep -E '^encoded' $OUT/startup_${b}_$k.log)" | tee -a $OUT/startup.txt
      done; done
      grep -E "x265hip-startup" $OUT/startup_hip_3.log | grep -v -E "device copy of a source|reference-picture mirror" | tee -a $OUT/startup.txt | tail -30
      grep -c "create: " $OUT/startup_hip_3.log | sed 's/^/creates: /' | tee -a $OUT/startup.txt
      grep -E "x265hip-startup" $OUT/startup_hip_3.log | head -3; grep -E "x265hip-startup" $OUT/startup_hip_3.log | grep -E "create" | sed -n '1p;$p' ;;
    cpuprofile)
      clip /tmp/bench240.yuv 240
      for k in 1 2 3 4 5 6 7 8 9 10; do
        X265HIP_CPUSAMPLE_OUT=/tmp/s$k.bin LD_PRELOAD=tools/prof/libcpusample.so integration/_build/x265_hip_8bit --input /tmp/bench240.yuv --input-res 1920x1080 --fps 30 --frames 240 --preset medium --me hex --pools 24 -F 6 -o /tmp/p.hevc 2>&1 | grep -E "encoded|x265hip" > $OUT/cpuprofile_run$k.log
        python tools/prof/resolve.py /tmp/s$k.bin 200 > $OUT/cpu_profile_run$k.txt 2>&1
      done
      python tools/prof/merge.py $OUT/cpu_profile_run*.txt > $OUT/cpu_profile_bound_encoder.txt
      head -48 $OUT/cpu_profile_bound_encoder.txt ;;
    *) echo "unknown task $task"; exit 2 ;;
  esac
done
