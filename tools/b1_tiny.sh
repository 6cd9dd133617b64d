#!/bin/bash
# batch 1: tiny tile (64 px x 64 couts, flavour 8) with few / no K slices vs the 128-px tile with maximal split-K (flavour 3); time includes the reduce launch
cd $GRAFT_REPO_ROOT/tools
run() { timeout 60 ./conv_bench.out "$@" 2>&1 | grep -v "check vs" | sed 's/TFLOP.*wgs/wgs/'; }
echo "== 64x64 192->192 (k3)";  run 1 64 64 192 192 9 0 96 3 3 2 0 0 1; for k in 1 2 3; do run 1 64 64 192 192 9 0 64 $k 8 2 0 0 1; done
echo "== 64x64 384->192 (k6)";  run 1 64 64 384 192 9 0 96 6 3 1; for k in 1 2 3; do run 1 64 64 384 192 9 0 64 $k 8 1; done
echo "== 64x64 384->384 (k6)";  run 1 64 64 384 384 9 0 96 4 3 1; for k in 1 2; do run 1 64 64 384 384 9 0 64 $k 8 1; done
echo "== 64x64 576->192 (k9)";  run 1 64 64 576 192 9 0 96 8 3 1; for k in 1 2 3; do run 1 64 64 576 192 9 0 64 $k 8 1; done
echo "== 32x32 384->384 (k6)";  run 1 32 32 384 384 9 0 96 6 3 2 0 0 1; for k in 1 2 3 6; do run 1 32 32 384 384 9 0 64 $k 8 2 0 0 1; done
echo "== 32x32 768->384 (k12)"; run 1 32 32 768 384 9 0 96 12 3 1; for k in 2 4 6; do run 1 32 32 768 384 9 0 64 $k 8 1; done
echo "== 16x16 576->576 (k9)";  run 1 16 16 576 576 9 0 96 9 3 1; for k in 3 6 9; do run 1 16 16 576 576 9 0 64 $k 8 1; done
echo "== 16x16 1344->576 (k21)"; run 1 16 16 1344 576 9 0 96 21 3 1; for k in 7 14 21; do run 1 16 16 1344 576 9 0 64 $k 8 1; done
