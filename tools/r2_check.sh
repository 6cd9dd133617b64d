#!/bin/bash
# round-2 check after the shared epilogue unit + upstream prefetch: conv micro-bench, targeted parity tests, cascade bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
cd tools
run() { timeout 60 "$@" 2>&1; }
for shape in "64 64 64 384 384 9 0 128" "64 64 64 192 192 9 0 96" "64 32 32 576 576 9 0 96"; do
  for epi in 1 2; do
    run ./conv_bench.out $shape 1 2 $epi; run ./conv_bench.out $shape 1 5 $epi
  done
done
cd ..
} > gpurun_out/r2_conv.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_edges.py tests/test_world_pipeline_gpu.py -x -q -m gpu > gpurun_out/r2_tests.txt 2>&1
timeout 600 python bench.py --workload cascade --steps 2 --warmup 1 > gpurun_out/bench_cascade2.json 2> gpurun_out/bench_cascade2.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_grid8_x.json 2> gpurun_out/bench_grid8_x.err
tail -3 gpurun_out/r2_tests.txt; cat gpurun_out/r2_conv.txt; cat gpurun_out/bench_cascade2.json | cut -c1-600; cat gpurun_out/bench_grid8_x.json | cut -c1-400
