#!/bin/bash
# round 6, experiment 7: the wide tile's LDS-DMA 1x1 tail (decoder blocks' conv_res1 = 3x3 conv + fused 1x1 skip conv) against conv_glds's DMA stream on single
# layers (TD_SEG2 = Cin of the 1x1 segment), then the engine tests, per-op table and bench A/B with / without tails on the wide tile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp7.txt; : > $O
run() { echo "## TD_SEG2=$S2 $*" >> $O; TD_SEG2=$S2 timeout 120 tools/conv_bench.out $* 2>&1 | grep -E "us  |check wide|error|Error" >> $O; }
for rep in 1 2; do
S2=576,1; run 64 64 64 192 192 9 0 96 1 3 2 0 0 1;  run 64 64 64 192 192 9 0 96 1 2 2 0 0 1; run 64 64 64 192 192 9 0 96 1 9 2 0 0 1     # dec.512 block0.conv_res1 (k3 + 9)
S2=384,1; run 64 64 64 192 192 9 0 96 1 3 2 0 0 1;  run 64 64 64 192 192 9 0 96 1 9 2 0 0 1                                                # dec.512 block1-3.conv_res1 (k3 + 6)
S2=768,1; run 64 32 32 384 384 9 0 128 1 3 2 0 0 1; run 64 32 32 384 384 9 0 96 1 9 2 0 0 1                                                # dec.256 block1-2.conv_res1 (k6 + 12)
S2=960,1; run 64 16 16 576 576 9 0 96 1 3 2 0 0 1;  run 64 16 16 576 576 9 0 96 1 9 2 0 0 1                                                # dec.128 (k9 + 15)
S2=128,1; run 4 512 512 64 64 9 0 64 1 3 2 0 0 1;   run 4 512 512 64 64 9 0 64 1 9 2 0 0 1                                                 # decoder model, 512x512 dec conv_res1
done
S2=192,1; run 3 40 24 192 192 9 1 96 1 9 2 0 0 1
echo "## pure 1x1 (no 3x3 part)" >> $O
timeout 120 tools/conv_bench.out 64 32 32 192 384 1 0 96 1 9 0 2>&1 | grep -E "us  |check wide" >> $O
echo "# engine tests" >> $O
timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py -x -q -m gpu -k "wide or batch64 or config2 or forward or decoder_window" -s > gpurun_out/r06_exp7_tests.txt 2>&1
grep -E "wide tile|launches on the wide|passed|failed|Error|error|assert" gpurun_out/r06_exp7_tests.txt | head -30 >> $O
for o in "glds_wide_tail=0" ""; do TD_OPTS="$o" TD_TOP=90 timeout 300 python tools/profile_ops.py 64 bf16 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_exp7_per_op_b64_${o:-default}.txt; echo "[$o] $(head -1 gpurun_out/r06_exp7_per_op_b64_${o:-default}.txt)" >> $O; done
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide_tail=0" "" >> $O 2>&1
AB_ROUNDS=1 tools/ab.sh bench --workload cascade -- "glds_wide_tail=0" "" >> $O 2>&1
cat $O
