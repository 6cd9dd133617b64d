#!/bin/bash
# conv_ps (persistent-stream, flavours 6 = 8-wave tiles, 7 = 4-wave tiles) vs conv_glds (2 / 3): time + bit-identity
cd $GRAFT_REPO_ROOT/tools
run() { timeout 120 ./conv_bench.out "$@" 2>&1; }
for shape in "64 64 64 192 192 9 0 96 1 F 2 0 0 1" "64 64 64 192 192 9 0 96 1 F 1 0 0 0" "64 64 64 384 192 9 0 96 1 F 1 0 0 0" "64 64 64 384 384 9 0 128 1 F 1 0 0 0" "64 32 32 384 384 9 0 128 1 F 2 0 0 1" "64 32 32 384 384 9 0 128 1 F 1 0 0 0" "64 16 16 576 576 9 0 96 1 F 1 0 0 0" "64 16 16 576 576 9 0 96 1 F 2 0 0 1" "64 8 8 768 768 9 0 96 1 F 1 0 0 0" "3 40 24 192 192 9 1 96 1 F 2 0 0 1"; do
  for f in 3 7 2 6; do run ${shape/F/$f}; done
done
echo "== chain, reversed second layer"
for f in 3 7; do for c in 1 2; do run 64 64 64 192 192 9 0 96 1 $f 2 0 $c 1; done; done
