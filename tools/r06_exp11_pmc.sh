#!/bin/bash
# round 6: PMC breakdown of the decoder model's 64 -> 64 layer at 512x512 on the wide tile (flavour 9) and its persistent / role-split instantiation (10)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export GRAFT_REPO_ROOT=$(pwd) TD_NO_CMP=1
O=gpurun_out/r06_exp11_pmc.txt; : > $O
for a in "4 512 512 64 64 9 0 64 1 9 1" "4 512 512 64 64 9 0 64 1 10 1" "4 512 512 64 64 9 0 64 1 9 0" "4 512 512 64 64 9 0 64 1 9 2 0 0 1"; do
  echo "## $a" >> $O
  ARGS="$a" bash tools/pmc_conv.sh > /dev/null 2>&1
  cat gpurun_out/pmc_conv/summary.txt >> $O
done
cat $O
