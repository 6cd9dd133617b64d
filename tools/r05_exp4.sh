#!/bin/bash
# round 5, experiment 4 (short): (A) the 128 px x 32 cout tile of conv_sb (mt 4) against 64 x 64 on the 64x64 / 32x32 levels of one and two tiles;
# (B) phase traces (s_memtime) of the conv_glds workgroup life, round-4 kernel against this tree; (C) the small-batch tests; (D) per-op batch 1 with
# sb_m4 on / off and the single-tile leg.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp4.txt; : > $O
echo "# (A) conv_sb tile: mt 4 nt 1 (128 px x 32 couts) vs mt 2 nt 2 (64 x 64); N H W Cin Cout mt nt Cin1x1 epi xform order" >> $O
sb() { echo "## $*" >> $O; timeout 120 tools/sb_bench.out $* 2>&1 | grep -v "conv_glds" | head -4 >> $O; }
for mtnt in "4 1" "2 2"; do
  sb 1 64 64 192 192 $mtnt 0 1 2 1
  sb 1 64 64 192 192 $mtnt 0 2 0 1 0 0 1
  sb 1 64 64 576 192 $mtnt 0 1 0 1
  sb 1 64 64 192 192 $mtnt 576 2 0 1
  sb 1 64 64 384 384 $mtnt 0 1 1 1 2
  sb 2 32 32 384 384 $mtnt 0 1 2 1
  sb 2 32 32 384 384 $mtnt 768 2 0 1
  sb 3 40 40 192 192 $mtnt 0 2 0 0 0 0 1
done
echo "# (B) conv_glds phase traces: base = round-4 kernel, new = this tree" >> $O
for L in "64 64 64 192 192 9 0 96 1 3 1" "64 64 64 192 192 9 0 96 1 3 2 0 0 1" "4 512 512 64 64 9 0 64 1 3 1" "4 512 512 64 64 9 0 64 1 3 2 0 0 1" "64 32 32 384 384 9 0 128 1 3 1"; do
  echo "## $L" >> $O
  for b in trace_base trace; do timeout 120 tools/conv_bench_$b.out $L 2>&1 | grep -E "us  |trace \(|taps per WG|of the epilogue|timeline" | sed "s/^/  $b: /" >> $O; done
done
echo "# (C) tests" >> $O
timeout 900 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_attention.py -x -q -m gpu > gpurun_out/r05_exp4_tests.txt 2>&1
tail -4 gpurun_out/r05_exp4_tests.txt >> $O
echo "# (D) per-op batch 1 / 2, sb_m4 on / off; single tile" >> $O
for o in "" "sb_m4=0"; do for n in 1 2; do TD_OPTS=$o TD_TOP=90 timeout 200 python tools/profile_ops.py $n bf16 2>/dev/null > gpurun_out/r05_exp4_per_op_b${n}_${o:-default}.txt; echo "[$o] $(head -1 gpurun_out/r05_exp4_per_op_b${n}_${o:-default}.txt)" >> $O; done; done
for o in "" "sb_m4=0" "" "sb_m4=0"; do echo "[single tile x 20 steps, $o]" >> $O; timeout 200 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts "$o" 2>/dev/null | cut -c120-230 >> $O; done
cat $O
