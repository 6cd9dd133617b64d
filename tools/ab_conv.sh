#!/bin/bash
# A/B on one box: tools/cb_head.out (conv kernels of the previous commit) vs tools/conv_bench.out (working tree), interleaved, 2 rounds
cd $GRAFT_REPO_ROOT/tools
shapes=(
 "64 64 64 384 384 9 0 128 1 2 1" "64 64 64 384 384 9 0 128 1 2 2"
 "64 64 64 192 192 9 0 96 1 2 1"  "64 64 64 192 192 9 0 96 1 2 2"
 "64 64 64 576 192 9 0 96 1 2 1"  "64 64 64 384 192 9 0 96 1 2 1"
 "64 32 32 576 576 9 0 96 1 2 1"  "64 32 32 576 576 9 0 96 1 2 2"
 "64 32 32 384 384 9 0 128 1 3 1" "64 32 32 384 384 9 0 128 1 3 2"
 "64 32 32 960 384 9 0 128 1 3 1" "64 32 32 768 384 9 0 128 1 3 1"
 "64 16 16 1344 576 9 0 96 1 3 1" "64 16 16 576 576 9 0 96 1 3 2"
 "64 8 8 768 768 9 0 96 1 3 2"
)
for r in 1 2; do
for s in "${shapes[@]}"; do
  a=$(timeout 60 ./cb_head.out $s 2>&1 | grep -o ": [0-9.]* us" | head -1)
  b=$(timeout 60 ./conv_bench.out $s 2>&1 | grep -o ": [0-9.]* us" | head -1)
  echo "$s | head$a | new$b"
done
done
