#!/bin/bash
# runs the conv_bench ablation binaries over a few layer shapes (args: N H W Cin Cout taps xform bn ksplit flavor epi)
cd $GRAFT_REPO_ROOT/tools
for shape in "64 64 64 384 384 9 0 128 1 2 1" "64 64 64 192 192 9 0 96 1 2 2" "64 32 32 576 576 9 0 96 1 2 1" "64 16 16 768 768 9 0 96 1 3 1"; do
  for b in base nobload nobstore noepi noall base; do
    echo -n "$b: "; timeout 60 ./cb_$b.out $shape
  done
done
