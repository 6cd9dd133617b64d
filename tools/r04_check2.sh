#!/bin/bash
# round-4 check after touching the throughput kernel's prologue: tile-variant bit-identity + parity tests, default bench, cascade bench, batch-1 bench
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_small_batch.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_tests2.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/r04_bench_grid8.txt 2>gpurun_out/r04_bench_grid8.err
python bench.py --workload tiles --tiles-per-step 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency > gpurun_out/r04_bench_b1.txt 2>&1
python bench.py --workload cascade --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_cascade.txt 2>gpurun_out/r04_bench_cascade.err
tail -3 gpurun_out/r04_tests2.txt; tail -1 gpurun_out/r04_bench_grid8.txt | cut -c1-900; tail -1 gpurun_out/r04_bench_b1.txt | cut -c1-200; tail -1 gpurun_out/r04_bench_cascade.txt | cut -c1-400
