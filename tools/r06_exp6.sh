#!/bin/bash
# round 6, experiment 6: what the wide tile's epilogue STORES cost, and what their pattern costs: normal build / no stores / the same bytes to wave-linear
# (wrong) addresses where every store instruction covers 1 KB of whole 128-byte lines instead of 32 runs of 32 bytes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp6.txt; : > $O
for rep in 1 2; do
for L in "64 64 64 192 192 9 0 96 1 9 1" "64 64 64 192 192 9 0 96 1 9 2 0 0 1" "64 64 64 576 192 9 0 96 1 9 1" "4 512 512 64 64 9 0 64 1 9 2 0 0 1"; do
  echo "## $L" >> $O
  for b in conv_bench conv_bench_e1 conv_bench_e2; do echo -n "  $b: " >> $O; TD_NO_CMP=1 timeout 120 tools/$b.out $L 2>&1 | grep -E "us  " >> $O; done
done; done
cat $O
