#!/bin/bash
# round 5, experiments 5 and 6 (short; the same script run on two builds): the residual-run addresses made in the prologue (5) and the runs requested at tap 6 of the last K-group instead of tap 0 (6): phase traces and bits.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp6.txt; : > $O
for L in "64 64 64 192 192 9 0 96 1 3 2 0 0 1" "64 64 64 192 192 9 0 96 1 3 2 0 0 0" "4 512 512 64 64 9 0 64 1 3 2 0 0 1" "64 32 32 384 384 9 0 128 1 3 2 0 0 1" "64 16 16 576 576 9 0 96 1 3 2 0 0 1" "64 64 64 384 384 9 0 128 1 2 2 0 0 1" "64 8 8 768 768 9 0 96 1 3 2 0 0 1" "3 20 20 192 192 9 0 96 1 3 2 0 0 1"; do
  echo "## $L" >> $O
  for b in trace_base trace; do timeout 120 tools/conv_bench_$b.out $L 2>&1 | grep -E "us  |trace \(" | sed "s/^/  $b: /" | cut -c1-330 >> $O; done
  for b in base new; do TD_DUMP=gpurun_out/cb_$b.bin timeout 120 tools/conv_bench_$b.out $L | head -1 | sed "s/^/  $b: /" >> $O; done
  cmp gpurun_out/cb_base.bin gpurun_out/cb_new.bin > /dev/null && echo "  bits: identical" >> $O || echo "  bits: DIFFER" >> $O
done
rm -f gpurun_out/cb_*.bin
timeout 600 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "tile_variants or dma_ragged or batch64" > gpurun_out/r05_exp6_tests.txt 2>&1; tail -3 gpurun_out/r05_exp6_tests.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 > gpurun_out/r05_exp6_per_op_batch64.txt 2>/dev/null; head -1 gpurun_out/r05_exp6_per_op_batch64.txt >> $O
cat $O
