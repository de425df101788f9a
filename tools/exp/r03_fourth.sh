#!/bin/bash
# Round 3: how many host cores does the box really give us, and where does the wall clock of a short encode go outside the encoder's own fps clock
set -u
OUT=gpurun_out/r03_d
mkdir -p $OUT
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; grep Cpus_allowed_list /proc/self/status; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8;
  grep -c processor /proc/cpuinfo; grep "model name" /proc/cpuinfo | head -1; cat /proc/loadavg; } | tee $OUT/host.txt
# a spin test: N busy processes for 2 s each, how much CPU time do they get in total?
python3 - <<'PY' | tee -a $OUT/host.txt
import multiprocessing as mp, time, os
def spin(q):
    t0 = time.process_time(); w0 = time.perf_counter()
    while time.perf_counter() - w0 < 2.0: pass
    q.put(time.process_time() - t0)
for n in (1, 8, 16, 32, 64, 128):
    q = mp.Queue(); ps = [mp.Process(target=spin, args=(q,)) for _ in range(n)]
    w0 = time.perf_counter()
    [p.start() for p in ps]; tot = sum(q.get() for _ in ps); [p.join() for p in ps]
    print("spin %3d processes: %.1f CPU-s in %.2f s wall -> %.1f cores" % (n, tot, time.perf_counter() - w0, tot / 2.0))
PY
python3 -c "
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/c.yuv', 1920, 1080, 120, seed=4321)"
R=$(pwd)/oracle/_ref
A="--input /tmp/c.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
for i in 1 2; do
  TIMEFORMAT="hip wall %R s user %U s sys %S s"
  { time X265HIP_DEBUG_STARTUP=1 X265HIP=require X265HIP_VERBOSE=1 $R/x265_hip_8bit $A -o /tmp/a.hevc 2> $OUT/startup_$i.log ; } 2>&1 | tee -a $OUT/startup_summary.txt
  grep "^encoded" $OUT/startup_$i.log | tee -a $OUT/startup_summary.txt
  grep "x265hip-startup" $OUT/startup_$i.log | grep -v "source picture" | head -12 | tee -a $OUT/startup_summary.txt
  grep "x265hip-startup" $OUT/startup_$i.log | grep "source picture" | head -4 | tee -a $OUT/startup_summary.txt
  grep "x265hip-startup" $OUT/startup_$i.log | tail -3 | tee -a $OUT/startup_summary.txt
done
TIMEFORMAT="ref wall %R s user %U s sys %S s"
{ time $R/x265_8bit $A -o /tmp/r.hevc 2> /tmp/r.log ; } 2>&1 | tee -a $OUT/startup_summary.txt
grep "^encoded" /tmp/r.log | tee -a $OUT/startup_summary.txt
TIMEFORMAT="hip --pools 16 wall %R s user %U s sys %S s"
{ time X265HIP=require $R/x265_hip_8bit $A --pools 16 -o /tmp/a.hevc 2> /tmp/p.log ; } 2>&1 | tee -a $OUT/startup_summary.txt
grep "^encoded" /tmp/p.log | tee -a $OUT/startup_summary.txt
TIMEFORMAT="ref --pools 16 wall %R s user %U s sys %S s"
{ time $R/x265_8bit $A --pools 16 -o /tmp/r.hevc 2> /tmp/r.log ; } 2>&1 | tee -a $OUT/startup_summary.txt
grep "^encoded" /tmp/r.log | tee -a $OUT/startup_summary.txt
