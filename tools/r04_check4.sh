#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_small_batch.py tests/test_gpu_attention.py -x -q -m gpu -k "variants or dma or small or arbitrary or tile_shape or ragged" 2>&1 | tail -4 > gpurun_out/r04_tests4.txt
for i in 1 2; do python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r.get('small_batch_kernel'))"; done
python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency 2>/dev/null | tail -1 | cut -c1-200
cat gpurun_out/r04_tests4.txt
