#!/bin/bash
# round 5, experiment 3 (short check before the final collection): conv_s16 restricted to the 16x16 level + sixteen-at-a-time 1/rms table + the fused
# qkv pack launch: tests that touch them, per-op tables at batch 1 (s16 on / off), the single-tile leg, the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp3.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_attention.py tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -m gpu -k "not third_order" > gpurun_out/r05_exp3_tests.txt 2>&1
tail -5 gpurun_out/r05_exp3_tests.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 1 bf16 > gpurun_out/r05_exp3_per_op_batch1.txt 2>/dev/null; head -1 gpurun_out/r05_exp3_per_op_batch1.txt >> $O
TD_OPTS=s16=0 TD_TOP=90 timeout 200 python tools/profile_ops.py 1 bf16 > gpurun_out/r05_exp3_per_op_batch1_s16off.txt 2>/dev/null; head -1 gpurun_out/r05_exp3_per_op_batch1_s16off.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 > gpurun_out/r05_exp3_per_op_batch64.txt 2>/dev/null; head -1 gpurun_out/r05_exp3_per_op_batch64.txt >> $O; grep attn gpurun_out/r05_exp3_per_op_batch64.txt >> $O
for o in "" "s16=0" "" "s16=0"; do echo "[single tile x 20 steps, $o]" >> $O; timeout 200 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts "$o" 2>/dev/null | cut -c1-230 >> $O; done
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_exp3_bench.json 2> gpurun_out/r05_exp3_bench.err
cut -c1-1500 gpurun_out/r05_exp3_bench.json >> $O
cat $O
