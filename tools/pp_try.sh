#!/bin/bash
# persistent ping-pong conv (flavor 5) vs LDS-DMA conv (flavor 2 big / 3 small): timing + bit-exactness
cd $GRAFT_REPO_ROOT/tools
run() { timeout 60 ./conv_bench.out "$@" 2>&1; }
for shape in "64 64 64 384 384 9 0 128" "64 64 64 192 192 9 0 96" "64 64 64 384 192 9 0 96" "64 32 32 576 576 9 0 96" "64 32 32 960 384 9 0 128" "64 16 16 768 768 9 0 96"; do
  for epi in 1 2; do
    run $shape 1 2 $epi; run $shape 1 5 $epi; run $shape 1 3 $epi
  done
done
# ragged sizes / small batch: correctness
run 3 40 24 128 128 9 1 128 1 5 1
run 2 16 48 192 96 9 0 96 1 5 2
