#!/bin/bash
# round 6, experiment 9: attention with the softmax's reference point and row sum on the matrix pipe (attn_fold, head dims <= 80) against the round-4 kernel:
# standalone harness (time, fp64 reference), phase trace, then the engine tests and the MFMA-busy report.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp9.txt; : > $O
for rep in 1 2; do
for L in "2 8 4096 4096 40" "2 8 4096 4096 64" "2 8 1024 1024 80" "2 8 4096 4096 128" "64 12 256 256 64" "64 12 64 64 64" "2 8 4096 77 40" "1 3 700 333 24" "2 8 4096 1000 40" "1 2 300 4096 8" "1 2 513 129 16"; do
  echo "## $L" >> $O
  for b in attn_bench_base attn_bench; do echo -n "  $b: " >> $O; timeout 120 tools/$b.out $L 2>&1 | grep -E "us  |rel-RMS" | tr '\n' ' ' >> $O; echo >> $O; done
done; done
echo "# phase traces (new kernel)" >> $O
for L in "2 8 4096 4096 40" "2 8 4096 4096 64"; do timeout 120 tools/attn_bench_trace.out $L 2>&1 | grep -v "^$" >> $O; done
echo "# tests" >> $O
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_parity.py -x -q -m gpu -k "attention or attn or forward or taps or block" > gpurun_out/r06_exp9_tests.txt 2>&1; tail -4 gpurun_out/r06_exp9_tests.txt >> $O
bash tools/attn_profile.sh > gpurun_out/attn_profile.log 2>&1
cat gpurun_out/attn_profile.txt >> $O 2>/dev/null
cat $O
