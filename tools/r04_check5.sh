#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_world_pipeline_gpu.py tests/test_gpu_bench_config.py -x -q -m gpu -k "cascade or resident or latent or stage or pipeline or world or config4 or replica or decoder" 2>&1 | tail -6 > gpurun_out/r04_tests5.txt
python bench.py --workload cascade --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r04_bench_cascade.err | tail -1 > gpurun_out/r04_bench_cascade.txt
cat gpurun_out/r04_tests5.txt; cut -c1-240 gpurun_out/r04_bench_cascade.txt; tail -3 gpurun_out/r04_bench_cascade.err | cut -c1-300
