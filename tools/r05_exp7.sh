#!/bin/bash
# round 5, experiment 7 (short): conv_glds with the patch-coordinate decode kept out of the segment loop's preheader (no more scratch in the bn 128 / bn 96 prologues,
# 25-30 fewer VGPRs): bits against the round-4 kernel, time; the 64-cout tile at three workgroups per CU again (40 bytes of scratch now instead of 132).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp7.txt; : > $O
ab() {
  echo "## $*" >> $O
  for b in base new; do
    timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
    TD_DUMP=gpurun_out/cb_$b.bin timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
  done
  cmp gpurun_out/cb_base.bin gpurun_out/cb_new.bin > /dev/null && echo "  bits: identical" >> $O || echo "  bits: DIFFER" >> $O
}
ab 64 32 32 384 384 9 0 128 1 3 1
ab 64 32 32 384 384 9 0 128 1 3 2 0 0 1
ab 64 64 64 384 384 9 0 128 1 2 2 0 0 1
ab 64 64 64 192 192 9 0 96 1 3 1
ab 64 64 64 192 192 9 0 96 1 3 2 0 0 1
ab 64 16 16 576 576 9 0 96 1 3 2 0 0 1
ab 64 8 8 768 768 9 0 128 1 2 1
ab 3 20 20 192 192 9 0 96 1 3 2 0 0 1
TD_SEG2=384,1 ab 64 32 32 384 384 9 0 128 1 3 0
TD_SEG2=384,1 ab 64 64 64 192 192 9 0 96 1 2 2
ab 64 32 32 192 384 1 0 128 1 3 0
echo "# decoder 512x512 level: new (two workgroups per CU) vs occ3 (three)" >> $O
for L in "4 512 512 64 64 9 0 64 1 3 1" "4 512 512 64 64 9 0 64 1 3 2 0 0 1" "4 512 512 128 64 9 0 64 1 3 1" "4 512 512 192 64 9 0 64 1 3 1" "64 64 64 192 64 9 0 64 1 3 0"; do
  echo "## $L" >> $O
  for b in new occ3 new occ3; do timeout 120 tools/conv_bench_$b.out $L | head -1 | sed "s/^/  $b: /" >> $O; done
done
for b in new occ3; do TD_DUMP=gpurun_out/cb_$b.bin timeout 120 tools/conv_bench_$b.out 4 512 512 64 64 9 0 64 1 3 2 0 0 1 > /dev/null; done
cmp gpurun_out/cb_new.bin gpurun_out/cb_occ3.bin > /dev/null && echo "  occ3 bits: identical" >> $O || echo "  occ3 bits: DIFFER" >> $O
rm -f gpurun_out/cb_*.bin
timeout 600 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "tile_variants or dma_ragged or batch64" > gpurun_out/r05_exp7_tests.txt 2>&1; tail -3 gpurun_out/r05_exp7_tests.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 > gpurun_out/r05_exp7_per_op_batch64.txt 2>/dev/null; head -1 gpurun_out/r05_exp7_per_op_batch64.txt >> $O
timeout 200 python tools/profile_model.py decoder 4 512 2>/dev/null | head -3 >> $O
cat $O
