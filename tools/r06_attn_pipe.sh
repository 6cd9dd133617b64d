#!/bin/bash
# round 6: the software-pipelined attention loop (TD_ATTN_PIPE=1, default) against the unpipelined one (TD_ATTN_PIPE=0) in the standalone harness: time, error against
# the fp64 reference, and the outputs compared bit for bit (TD_ATTN_PIPE_MIN=0 TD_ATTN_PIPE_MAX_WGS=huge: the pipelined loop on every shape, also where the launcher would not
# pick it); then the attention tests and, with "profile", the MFMA-busy report.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_attn_pipe.txt; : > $O
B=tools/attn_bench.out
for shape in "2 8 4096 4096 40" "2 8 4096 4096 64" "2 8 4096 4096 128" "2 8 4096 4096 32" "2 8 4096 4096 96" "2 8 4096 4096 80" "2 8 4096 4096 8" "2 8 4096 4096 120" "2 8 1024 1024 64" "2 8 1024 1024 80" "64 12 256 256 64" "64 12 64 64 64" "2 8 1000 777 40" "2 8 4096 77 40" "1 1 5 3 24" "1 2 700 513 72" "1 2 300 640 128"; do
  for rep in 1 2; do
  for p in 0 1; do
    echo "## TD_ATTN_PIPE=$p $shape" >> $O
    TD_ATTN_PIPE_MIN=0 TD_ATTN_PIPE_MAX_WGS=100000000 TD_ATTN_PIPE=$p TD_ATTN_DUMP=/tmp/attn_d$p.bin timeout 120 $B $shape 50 >> $O 2>&1
  done; done
  cmp /tmp/attn_d0.bin /tmp/attn_d1.bin >> $O 2>&1 && echo "  outputs IDENTICAL" >> $O
done
if [ "$1" != "notest" ]; then echo "# tests" >> $O; timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -5 >> $O; fi
if [ "$1" = "profile" ]; then bash tools/attn_profile.sh > gpurun_out/attn_profile.log 2>&1; cat gpurun_out/attn_profile.txt >> $O; fi
cat $O
