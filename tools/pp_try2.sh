#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
run() { timeout 60 "$@" 2>&1; }
for shape in "64 64 64 384 384 9 0 128" "64 64 64 192 192 9 0 96" "64 32 32 576 576 9 0 96" "64 16 16 768 768 9 0 96"; do
  for epi in 1 2; do
    run ./conv_bench.out $shape 1 2 $epi; run ./conv_bench.out $shape 1 5 $epi; run ./cbp_nodmap.out $shape 1 5 $epi | grep -v check; run ./cbp_noepi.out $shape 1 5 $epi | grep -v check
  done
done
run ./cbp_trace.out 64 64 64 384 384 9 0 128 1 5 1 | grep -v check
run ./cbp_trace.out 64 64 64 192 192 9 0 96 1 5 2 | grep -v check
run ./conv_bench.out 3 40 24 128 128 9 0 128 1 5 1
run ./conv_bench.out 2 16 48 192 96 9 0 96 1 5 2
run ./conv_bench.out 5 48 80 192 192 9 1 96 1 5 2
