#!/bin/bash
# round-4 quick check on the GPU box: small-batch tests, per-op table at batch 1, single-tile latency
mkdir -p gpurun_out
python -m pytest tests/test_gpu_small_batch.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r04_sb_tests.txt
python tools/profile_ops.py 1 bf16 > gpurun_out/r04_per_op_b1.txt 2>&1
TD_TOP=100 TD_OPTS=sb_prefetch=0 python tools/profile_ops.py 1 bf16 > gpurun_out/r04_per_op_b1_sb0.txt 2>&1
python bench.py --workload tiles --tiles-per-step 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency > gpurun_out/r04_bench_b1.txt 2>&1
python bench.py --workload tiles --tiles-per-step 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts sb_prefetch=0 > gpurun_out/r04_bench_b1_sb0.txt 2>&1
tail -3 gpurun_out/r04_sb_tests.txt; head -3 gpurun_out/r04_per_op_b1.txt; tail -1 gpurun_out/r04_bench_b1.txt | cut -c1-300; tail -1 gpurun_out/r04_bench_b1_sb0.txt | cut -c1-300
