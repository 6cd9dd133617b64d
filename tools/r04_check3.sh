#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_attention.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_tests3.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py -x -q -m gpu -k "shard or rank or grid32 or stream or replica" 2>&1 | tail -5 >> gpurun_out/r04_tests3.txt
python bench.py --workload cascade --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_cascade.txt 2>gpurun_out/r04_bench_cascade.err
cat gpurun_out/r04_tests3.txt; tail -1 gpurun_out/r04_bench_cascade.txt | cut -c1-300; tail -3 gpurun_out/r04_bench_cascade.err
