#!/bin/bash
# enqueue-only cascade: the tests that cover the changed entry points, then the cascade bench synchronous vs enqueue-only (interleaved twice)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_world_pipeline_gpu.py tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/cascade_tests.txt
cat gpurun_out/cascade_tests.txt
for r in 1 2; do
  for m in 1 0; do
    echo -n "cascade-sync $m: "
    timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 --cascade-sync $m 2>gpurun_out/cascade_mode_$m.err | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
  done
done | tee gpurun_out/r04_cascade_sync_vs_enqueue.txt
tail -3 gpurun_out/cascade_mode_0.err | cut -c1-300
