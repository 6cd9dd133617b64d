#!/bin/bash
# One decoder conv_res1 launch in isolation (3x3 segment + fused 1x1 skip segment) through conv_bench: register-staged vs LDS-DMA 1x1 path, timing
# and bit-for-bit comparison.  Written at the end of round 3 (after the GPU budget was spent): compiled, NOT yet run on hardware.
cd $GRAFT_REPO_ROOT/tools
run() { for d in 0 1; do echo -n "dma1x1=$d  "; TD_SEG2=$SEG2 TD_DMA1X1=$d timeout 60 ./conv_bench.out "$@" 2>&1 | sed 's/(.*of 2500)//'; done; }
echo "== dec.512x512_block1.conv_res1: 64x64, 192 ch 3x3 + 384 ch 1x1 -> 192, 8-wave tile bn96";  SEG2=384,1 run 64 64 64 192 192 9 0 96 1 2 2
echo "== dec.512x512_block0.conv_res1: 64x64, 192 ch 3x3 + 576 ch 1x1 -> 192";                     SEG2=576,1 run 64 64 64 192 192 9 0 96 1 2 2
echo "== dec.256x256_block1.conv_res1: 32x32, 384 ch 3x3 + 768 ch 1x1 -> 384, 4-wave tile bn128";  SEG2=768,1 run 64 32 32 384 384 9 0 128 1 3 2
echo "== dec.128x128_block1.conv_res1: 16x16, 576 ch 3x3 + 1152 ch 1x1 -> 576, 4-wave tile bn96";  SEG2=1152,1 run 64 16 16 576 576 9 0 96 1 3 2
echo "== dec.64x64_block1.conv_res1: 8x8, 768 ch 3x3 + 1536 ch 1x1 -> 768, narrow tile, ks2";      SEG2=1536,1 run 64 8 8 768 768 9 0 96 2 3 2
