#!/bin/bash
# round 5, experiment 1 (one gpurun call): (A) conv_glds epilogue rework (modulation rows through LDS, residual runs requested during the last K-group,
# straight-line unit loop) against the round-4 kernel, with bit comparison of out / out2 / sumsq; (B) the decoder's 64-channel 512x512 level: tile
# variants, three workgroups per CU, the persistent ping-pong flavour at bn 64; (C) a first pass of the GPU tests on the rebuilt library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp1.txt; : > $O
ab() {  # args: env-prefix, conv_bench args
  local envp="$1"; shift
  echo "## $envp $*" >> $O
  for b in base new; do
    env $envp timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
    env $envp TD_DUMP=gpurun_out/cb_$b.bin timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
  done
  cmp gpurun_out/cb_base.bin gpurun_out/cb_new.bin > /dev/null && echo "  bits: identical ($(stat -c %s gpurun_out/cb_new.bin) bytes)" >> $O || echo "  bits: DIFFER" >> $O
}
echo "# (A) epilogue rework: base = round-4 kernel, new = this tree" >> $O
ab X=1 64 64 64 192 192 9 0 96 1 3 1
ab X=1 64 64 64 192 192 9 0 96 1 3 2 0 0 1
ab X=1 64 64 64 192 192 9 0 96 1 3 2 0 0 0
ab X=1 64 64 64 384 384 9 0 128 1 3 2 0 0 1
ab X=1 64 64 64 384 192 9 0 96 1 3 1
ab X=1 64 64 64 384 384 9 0 128 1 2 1
ab X=1 64 64 64 384 384 9 0 128 1 2 2 0 0 1
ab X=1 64 32 32 384 384 9 0 128 1 3 1
ab X=1 64 32 32 384 384 9 0 128 1 3 2 0 0 1
ab X=1 64 16 16 576 576 9 0 96 1 3 2 0 0 1
ab X=1 64 16 16 576 576 9 0 96 1 3 1
ab X=1 64 8 8 768 768 9 0 96 1 3 1
ab X=1 64 8 8 768 768 9 0 96 1 3 2 0 0 1
ab X=1 64 8 8 768 768 9 0 128 1 2 1
ab X=1 64 8 8 768 768 9 0 128 1 2 2 0 0 1
ab X=1 64 8 8 768 768 1 0 96 1 3 2
ab TD_SEG2=384,1 64 64 64 192 192 9 0 96 1 3 2
ab "TD_SEG2=384,1 TD_DMA1X1=0" 64 64 64 192 192 9 0 96 1 3 2
ab X=1 3 20 20 192 192 9 0 96 1 3 2 0 0 1
ab X=1 3 20 20 192 192 9 0 96 1 3 1
ab X=1 4 512 512 64 64 9 0 64 1 3 1
ab X=1 4 512 512 64 64 9 0 64 1 3 2 0 0 1
echo "# (B) decoder 512x512 level, 64 channels: fl3 = small tile, fl2 = big tile, occ3 = small tile three workgroups per CU, fl5 = ping-pong bn 64" >> $O
for L in "4 512 512 64 64 9 0 64 1 3 1" "4 512 512 64 64 9 0 64 1 3 2 0 0 1" "4 512 512 128 64 9 0 64 1 3 1" "4 512 512 192 64 9 0 64 1 3 1" "4 512 512 64 64 9 0 64 1 2 1" "4 512 512 64 64 9 0 64 1 2 2 0 0 1"; do
  echo "## $L" >> $O
  for b in new occ3; do for r in 1 2; do timeout 120 tools/conv_bench_$b.out $L | head -1 | sed "s/^/  $b: /" >> $O; done; done
done
for L in "4 512 512 64 64 9 0 64 1 5 1" "4 512 512 64 64 9 0 64 1 5 2 0 0 1" "4 512 512 128 64 9 0 64 1 5 1" "4 512 512 192 64 9 0 64 1 5 1"; do
  echo "## $L" >> $O
  for r in 1 2; do timeout 120 tools/conv_bench_new.out $L | head -2 | sed "s/^/  new: /" >> $O; done
done
echo "## TD_SEG2=128,1 4 512 512 64 64 9 0 64 1 3 2 (dec conv_res1: 3x3 + fused 1x1 skip)" >> $O
for b in new occ3; do for r in 1 2; do TD_SEG2=128,1 timeout 120 tools/conv_bench_$b.out 4 512 512 64 64 9 0 64 1 3 2 | head -2 | sed "s/^/  $b: /" >> $O; done; done
rm -f gpurun_out/cb_*.bin
echo "# (C) GPU tests on the rebuilt library" >> $O
timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_small_batch.py tests/test_gpu_edges.py -x -q -m gpu -k "not config4 and not config3 and not third_order" > gpurun_out/r05_exp1_tests.txt 2>&1
tail -5 gpurun_out/r05_exp1_tests.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 > gpurun_out/r05_exp1_per_op_batch64.txt 2>/dev/null
head -3 gpurun_out/r05_exp1_per_op_batch64.txt >> $O
echo "# (D) decoder forward at batch 4, 512x512: producer-side activation on / off (the 64-channel level is HBM-bound: the second output is a quarter of its bytes)" >> $O
for o in "" "producer_act=0"; do echo "[$o]" >> $O; TD_OPTS=$o timeout 200 python tools/profile_model.py decoder 4 512 2>/dev/null | head -16 >> $O; done
echo "# (E) base forward at batch 64, 8x8 level on the small-batch flavour: 64 x 64 tiles (768 workgroups, 1.5 rounds) vs 64 x 32 (1536)" >> $O
TD_OPTS=sb_nt=1 TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 2>/dev/null | grep -E "kernel time|8x8" | head -30 >> $O
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/r05_exp1_bench.json 2> gpurun_out/r05_exp1_bench.err
cut -c1-400 gpurun_out/r05_exp1_bench.json >> $O
cat $O | tail -150
