#!/bin/bash
# phase traces (tools/sb_trace.out = sb_bench.hip built with -DTD_TRACE) of representative batch-1 layers;  $1 = deep (0/1)
B=tools/sb_trace.out; DEEP=${1:-0}
run() { echo "--- $*"; timeout 120 $B "$@" | grep -v "vs per-tap\|sumsq\|out2" || echo "FAILED($?) $*"; }
run 1 64 64 192 192 2 2 0 1 2
run 1 64 64 192 192 2 2 0 2 0
run 1 64 64 576 192 2 2 0 1 0
run 1 64 64 192 192 2 2 576 2 0
run 1 64 64 384 384 2 2 0 2 0
run 1 32 32 384 384 2 1 0 1 2
run 1 32 32 768 384 2 1 0 1 0
run 1 16 16 576 576 1 1 0 1 2
run 1 16 16 1152 576 1 1 0 1 0
run 1 8 8 768 768 1 1 0 1 2
run 1 8 8 1536 768 2 1 0 1 0
run 1 8 8 0 2304 2 1 768 0 0
