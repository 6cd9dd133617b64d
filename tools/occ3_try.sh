#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for shape in "64 64 64 192 192 9 0 BN 1 3 2 0 0 1" "64 64 64 192 192 9 0 BN 1 3 1 0 0 0" "64 32 32 384 384 9 0 BN 1 3 1 0 0 0" "64 16 16 576 576 9 0 BN 1 3 1 0 0 0"; do
  echo -n "bn96/128 2-per-CU: "; timeout 60 ./conv_bench.out ${shape/BN/96}
  echo -n "bn64     2-per-CU: "; timeout 60 ./conv_bench.out ${shape/BN/64}
  echo -n "bn64     3-per-CU: "; timeout 60 ./cb_occ3.out ${shape/BN/64}
done
