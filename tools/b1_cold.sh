#!/bin/bash
# batch 1: how much of the in-network per-layer time is cold weights/activations?  hot loop vs every launch after a 1 GiB cache flush
# (TD_COLD=1), and with the weights (2) / weights + activations (3) pulled back through the memory-side cache by a prefetch kernel
cd $GRAFT_REPO_ROOT/tools
run() { for m in 1 2 3; do TD_COLD=$m timeout 60 ./conv_bench.out "$@" 2>&1 | grep -v "check vs" | sed 's/TFLOP.*wgs/wgs/' | if [ $m = 1 ]; then cat; else grep cold; fi; done; }
echo "== 64x64 384->192 (k6): 128-px tile ks4"; run 1 64 64 384 192 9 0 96 4 3 1
echo "== 64x64 192->192 (k3)"; run 1 64 64 192 192 9 0 96 3 3 2 0 0 1
echo "== 32x32 768->384 (k12)"; run 1 32 32 768 384 9 0 96 12 3 1
echo "== 16x16 1344->576 (k21)"; run 1 16 16 1344 576 9 0 96 21 3 1
echo "== 8x8 1536->768 (k24)"; run 1 8 8 1536 768 9 0 96 24 3 1
