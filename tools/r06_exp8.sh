#!/bin/bash
# round 6, experiment 8: planner thresholds around the launches that stay on conv_glds (1x1 tails): the 4-wave tile for longer K loops; wide-tile grid threshold.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp8.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "decoder_window or wide" > gpurun_out/r06_exp8_tests.txt 2>&1; tail -3 gpurun_out/r06_exp8_tests.txt >> $O
for o in "" "glds_small_max_groups=12" "glds_small_max_groups=21"; do TD_OPTS="$o" TD_TOP=90 timeout 300 python tools/profile_ops.py 64 bf16 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_exp8_per_op_${o:-default}.txt; echo "[$o] $(head -1 gpurun_out/r06_exp8_per_op_${o:-default}.txt)" >> $O; done
AB_ROUNDS=2 tools/ab.sh bench -- "" "glds_small_max_groups=12" "glds_small_max_groups=21" "glds_wide_min_wgs=256" >> $O 2>&1
cat $O
