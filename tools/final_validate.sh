#!/bin/bash
# round-end validation on one MI355X: GPU test suite, smoke, default bench (with cpu_baseline), cascade bench
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/final_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final_tests.txt; fi   # SKIP_TESTS=1: a second collection of the SAME build on another box keeps the first one's test record
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/final_bench_grid8.json 2> gpurun_out/final_bench_grid8.err
timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 > gpurun_out/final_bench_cascade.json 2> gpurun_out/final_bench_cascade.err
timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 --cascade-sync 1 --no-cpu-baseline > gpurun_out/final_bench_cascade_sync.json 2> gpurun_out/final_bench_cascade_sync.err
timeout 600 python bench.py --workload cascade --dtype fp16 --steps 3 --warmup 1 > gpurun_out/final_bench_cascade_fp16.json 2> gpurun_out/final_bench_cascade_fp16.err
timeout 600 python bench.py --workload tiles --steps 5 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/final_bench_tiles.json 2> gpurun_out/final_bench_tiles.err
timeout 600 python bench.py --dtype fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/final_bench_grid8_fp16.json 2> gpurun_out/final_bench_grid8_fp16.err
TD_TOP=90 timeout 120 python tools/profile_ops.py 64 bf16 > gpurun_out/final_per_op_batch64.txt 2>/dev/null
TD_TOP=90 timeout 120 python tools/profile_ops.py 1 bf16 > gpurun_out/final_per_op_batch1.txt 2>/dev/null
timeout 600 python -c "from terrain_diffusion_amd.latency import measure_latency as m; import json; print(json.dumps(m(num_runs=60, dtype='bf16')))" 2>/dev/null | tail -1 > gpurun_out/final_ttft_ttst.json
timeout 300 bash tools/b1_timeline.sh > gpurun_out/final_b1_timeline.log 2>&1
timeout 600 bash tools/sb_layers.sh > gpurun_out/final_sb_layers.txt 2>&1
tail -4 gpurun_out/final_tests.txt; tail -2 gpurun_out/final_smoke.txt; cut -c1-700 gpurun_out/final_bench_grid8.json; cut -c1-300 gpurun_out/final_bench_cascade.json
