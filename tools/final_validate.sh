#!/bin/bash
# round-end validation on one MI355X: GPU test suite, smoke, default bench (with cpu_baseline), cascade bench
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/final_bench_grid8.json 2> gpurun_out/final_bench_grid8.err
timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 > gpurun_out/final_bench_cascade.json 2> gpurun_out/final_bench_cascade.err
tail -4 gpurun_out/final_tests.txt; tail -2 gpurun_out/final_smoke.txt; cut -c1-700 gpurun_out/final_bench_grid8.json; cut -c1-300 gpurun_out/final_bench_cascade.json
