#!/bin/bash
# round 6, experiment 5: the wide tile with the four-slot ring (half tiles requested three half-steps ahead) -- single layers against conv_glds, phase traces,
# its tests, per-op table and the bench A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp5.txt; : > $O
run() { echo "## $*" >> $O; timeout 120 tools/conv_bench.out $* 2>&1 | grep -E "us  |check|error|Error" >> $O; }
for rep in 1 2; do
run 64 64 64 192 192 9 0 96 1 3 1;            run 64 64 64 192 192 9 0 96 1 9 1
run 64 64 64 192 192 9 1 96 1 3 2 0 0 1;      run 64 64 64 192 192 9 1 96 1 9 2 0 0 1
run 64 64 64 384 192 9 0 96 1 3 2 0 0 1;      run 64 64 64 384 192 9 0 96 1 9 2 0 0 1
run 64 64 64 576 192 9 0 96 1 2 1;            run 64 64 64 576 192 9 0 96 1 9 1
run 64 64 64 384 384 9 0 128 1 3 1;           run 64 64 64 384 384 9 0 96 1 9 1
run 64 32 32 768 384 9 0 128 1 3 2 0 0 1;     run 64 32 32 768 384 9 0 96 1 9 2 0 0 1
run 64 16 16 576 576 9 0 96 1 3 1;            run 64 16 16 576 576 9 0 96 1 9 1
run 4 512 512 64 64 9 0 64 1 3 1;             run 4 512 512 64 64 9 0 64 1 9 1
run 4 512 512 64 64 9 0 64 1 3 2 0 0 1;       run 4 512 512 64 64 9 0 64 1 9 2 0 0 1
done
run 3 40 24 192 192 9 1 96 1 9 2 0 0 1
TD_SEG2=384,1 run 64 64 64 192 192 9 0 96 1 9 2 0 0 1
TD_SEG2=192,9 run 64 64 64 192 192 9 0 96 1 9 1
echo "# phase traces ('restage' column of flavour 9 = the vmcnt share of the tap-entry wait)" >> $O
for L in "64 64 64 192 192 9 0 96 1 9 1" "64 64 64 576 192 9 0 96 1 9 1" "4 512 512 64 64 9 0 64 1 9 1"; do
  echo "## $L" >> $O; TD_NO_CMP=1 timeout 120 tools/conv_bench_trace.out $L 2>&1 | grep -E "us  |trace \(|taps per WG|shader clock" >> $O
done
echo "# engine tests" >> $O
timeout 1200 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "wide or batch64 or config2" -s > gpurun_out/r06_exp5_tests.txt 2>&1
grep -E "wide tile|launches on the wide|passed|failed|Error|error|assert" gpurun_out/r06_exp5_tests.txt | head -30 >> $O
for o in "glds_wide=0" ""; do TD_OPTS="$o" TD_TOP=90 timeout 300 python tools/profile_ops.py 64 bf16 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_exp5_per_op_b64_${o:-default}.txt; echo "[$o] $(head -1 gpurun_out/r06_exp5_per_op_b64_${o:-default}.txt)" >> $O; done
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide=0" "" >> $O 2>&1
AB_ROUNDS=1 tools/ab.sh bench --workload cascade -- "glds_wide=0" "" >> $O 2>&1
cat $O
