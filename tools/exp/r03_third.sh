#!/bin/bash
# Round 3: the new bench line (N = 1, and N = 2 on one device over gloo), and how the host scales with N encoders side by side
set -u
OUT=gpurun_out/r03_c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 600 $OUT/bench_n1.err
python3 - $OUT/bench_n1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d.pop("frame_pass", None)
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "gpu_duty_cycle", "cpu_baseline")}, indent=1)[:5000])
for r in d["rooflines"]:
    print(r.get("kernel", r)[:100], r.get("frac"), r.get("launch_ms"))
PY
X265HIP_BENCH_SAME_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --no-frame-pass > $OUT/bench_n2same.json 2> $OUT/bench_n2same.err
tail -c 400 $OUT/bench_n2same.err
python3 - $OUT/bench_n2same.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "cpu_baseline")}, indent=1)[:3000])
except Exception as e:
    print("no line:", e)
PY
# host scaling: N encoders side by side on the one GPU, the pool split N ways
R=$(pwd)/oracle/_ref
A="--input /tmp/hs.yuv --input-res 1920x1080 --fps 30 --frames 120 --preset medium --me hex --hash 1"
python3 -c "
import sys; sys.path.insert(0, '.')
from x265_amd.synth import make_clip
make_clip('/tmp/hs.yuv', 1920, 1080, 120, seed=4321)"
for exe in x265_hip_8bit x265_8bit; do
for N in 1 2 4 8; do
  P=$((256 / N))
  t0=$(date +%s.%N)
  for i in $(seq $N); do
    X265HIP=require $R/$exe $A --pools $P -o /tmp/hs_$i.hevc > /tmp/hs_$i.log 2>&1 &
  done
  wait
  t1=$(date +%s.%N)
  echo "$exe N=$N pools=$P wall $(echo "$t1 - $t0" | bc) s -> $(echo "$N * 120 / ($t1 - $t0)" | bc -l | cut -c1-6) fps total; cli: $(grep -h '^encoded' /tmp/hs_*.log | sed 's/.*(\(.*\) fps).*/\1/' | tr '\n' ' ')" | tee -a $OUT/hostscale.txt
  rm -f /tmp/hs_*.log
done
done
