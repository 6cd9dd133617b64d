#!/bin/bash
# round 6, experiment 10: the wide tile's PERSISTENT tile loop (flavour 10) against one tile per workgroup (flavour 9) on single layers: the decoder model's
# 512x512 / 256x256 levels, the base model's 64x64 / 32x32 levels, ragged round counts; every persistent run is compared bit for bit with flavour 9.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp10.txt; : > $O
run() { echo "## $*" >> $O; timeout 120 tools/conv_bench.out $* 2>&1 | grep -E "us  |check persistent|check wide|error|Error" >> $O; }
for rep in 1 2; do
for f in 9 10; do
run 4 512 512 64 64 9 0 64 1 $f 1
run 4 512 512 64 64 9 0 64 1 $f 2 0 0 1
run 4 512 512 64 64 9 0 64 1 $f 0
run 4 512 512 128 64 9 0 64 1 $f 1
run 4 256 256 128 128 9 0 64 1 $f 1
run 64 64 64 192 192 9 0 64 1 $f 1
done
done
for f in 9 10; do
run 3 512 512 64 64 9 0 64 1 $f 1
run 5 512 512 64 64 9 0 64 1 $f 2 0 0 1
run 1 512 512 64 64 9 0 64 1 $f 1
run 4 512 512 192 64 9 0 64 1 $f 1
run 64 64 64 192 192 9 0 96 1 $f 1
run 64 64 64 192 192 9 0 96 1 $f 2 0 0 1
run 64 64 64 576 192 9 0 96 1 $f 1
run 64 32 32 384 384 9 0 96 1 $f 1
done
TD_PERSIST=256 run 4 512 512 64 64 9 0 64 1 10 1
TD_PERSIST=1024 run 4 512 512 64 64 9 0 64 1 10 1
TD_PERSIST=504 run 4 512 512 64 64 9 0 64 1 10 1
cat $O
