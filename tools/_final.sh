mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/gputests6.log 2>&1; tail -4 gpurun_out/gputests6.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_v4.json 2> gpurun_out/bench_r02_v4.err; tail -c 300 gpurun_out/bench_r02_v4.err
timeout 900 bash tools/collect_profiles.sh r02_v3 > gpurun_out/collect_r02_v3.log 2>&1
timeout 400 python tools/fuzz_encoder.py --seeds 200-239 --gpu --bits 10 --json gpurun_out/fuzz_gpu_main10_200_239.json > gpurun_out/fuzz_gpu_main10.log 2>&1; tail -1 gpurun_out/fuzz_gpu_main10.log
