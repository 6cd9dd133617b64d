#!/bin/bash
# the three cascade bench lines again (the per-resolution HBM rates of the first collection were lost to a label-format change in cascade_bench.py; same library build)
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 > gpurun_out/final_bench_cascade.json 2> gpurun_out/final_bench_cascade.err
timeout 600 python bench.py --workload cascade --steps 3 --warmup 1 --cascade-sync 1 --no-cpu-baseline > gpurun_out/final_bench_cascade_sync.json 2> gpurun_out/final_bench_cascade_sync.err
timeout 600 python bench.py --workload cascade --dtype fp16 --steps 3 --warmup 1 > gpurun_out/final_bench_cascade_fp16.json 2> gpurun_out/final_bench_cascade_fp16.err
timeout 200 python tools/profile_model.py decoder 4 512 > gpurun_out/final_decoder_forward_batch4.txt 2>/dev/null
cut -c1-200 gpurun_out/final_bench_cascade.json; head -3 gpurun_out/final_decoder_forward_batch4.txt
