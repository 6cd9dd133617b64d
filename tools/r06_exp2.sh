#!/bin/bash
# round 6, experiment 2: the wide tile (conv_glds_wide.hip, conv_bench flavour 9) against the planner's conv_glds tile on single layers (hot loop, 20 launches),
# with the output comparison (another K order: last-place differences expected), then phase traces of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp2.txt; : > $O
run() { echo "## $*" >> $O; timeout 120 tools/conv_bench.out $* 2>&1 | grep -E "us  |check|error|Error" >> $O; }
#   N H W Cin Cout taps xform bn ks flavour epi stagger chain out2
for rep in 1 2; do
run 64 64 64 192 192 9 0 96 1 3 1;            run 64 64 64 192 192 9 0 96 1 9 1
run 64 64 64 192 192 9 0 96 1 3 2 0 0 1;      run 64 64 64 192 192 9 0 96 1 9 2 0 0 1
run 64 64 64 192 192 9 1 96 1 3 2 0 0 1;      run 64 64 64 192 192 9 1 96 1 9 2 0 0 1
run 64 64 64 384 192 9 0 96 1 3 2 0 0 1;      run 64 64 64 384 192 9 0 96 1 9 2 0 0 1
run 64 64 64 576 192 9 0 96 1 2 1;            run 64 64 64 576 192 9 0 96 1 9 1
run 64 64 64 384 384 9 0 128 1 3 1;           run 64 64 64 384 384 9 0 96 1 9 1
run 64 32 32 384 384 9 0 128 1 3 1;           run 64 32 32 384 384 9 0 96 1 9 1
run 64 32 32 768 384 9 0 128 1 3 2 0 0 1;     run 64 32 32 768 384 9 0 96 1 9 2 0 0 1
run 4 512 512 64 64 9 0 64 1 3 1;             run 4 512 512 64 64 9 0 64 1 9 1
run 4 512 512 64 64 9 0 64 1 3 2 0 0 1;       run 4 512 512 64 64 9 0 64 1 9 2 0 0 1
run 4 256 256 128 128 9 0 128 1 3 1;          run 4 256 256 128 128 9 0 64 1 9 1
run 3 40 24 192 192 9 1 96 1 3 2 0 0 1;       run 3 40 24 192 192 9 1 96 1 9 2 0 0 1
done
echo "# 3x3 + 1x1 tail (TD_SEG2): correctness of the wide tile's (unpipelined) 1x1 path" >> $O
TD_SEG2=384,1 run 64 64 64 192 192 9 0 96 1 9 2 0 0 1
TD_SEG2=192,9 run 64 64 64 192 192 9 0 96 1 9 1
if [ -x tools/conv_bench_trace.out ]; then
  echo "# phase traces" >> $O
  for L in "64 64 64 192 192 9 0 96 1 3 1" "64 64 64 192 192 9 0 96 1 9 1" "64 64 64 192 192 9 0 96 1 3 2 0 0 1" "64 64 64 192 192 9 0 96 1 9 2 0 0 1" "4 512 512 64 64 9 0 64 1 3 1" "4 512 512 64 64 9 0 64 1 9 1"; do
    echo "## $L" >> $O; TD_NO_CMP=1 timeout 120 tools/conv_bench_trace.out $L 2>&1 | grep -E "us  |trace \(|taps per WG|of the epilogue|timeline|shader clock" >> $O
  done
fi
cat $O
echo "# engine: tests of the wide tile and of the tile-shape identities" >> $O
timeout 1200 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "wide or tile_variants or batch64 or config2" -s > gpurun_out/r06_exp2_tests.txt 2>&1
grep -E "wide tile|launches on the wide|passed|failed|Error|error|assert" gpurun_out/r06_exp2_tests.txt | head -30 >> $O
echo "# engine: per-op tables at batch 64, wide tile off / on" >> $O
for o in "glds_wide=0" ""; do TD_OPTS="$o" TD_TOP=90 timeout 300 python tools/profile_ops.py 64 bf16 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_exp2_per_op_b64_${o:-default}.txt; echo "[$o] $(head -1 gpurun_out/r06_exp2_per_op_b64_${o:-default}.txt)" >> $O; done
echo "# engine: bench A/B (grid8), wide tile off / on, dual_stream default" >> $O
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide=0" "" >> $O 2>&1
echo "# engine: bench A/B single lane" >> $O
AB_ROUNDS=1 tools/ab.sh bench -- "glds_wide=0,dual_stream=0" "dual_stream=0" >> $O 2>&1
cat $O
