#!/bin/bash
# round 3: chip-wide ramp of the first-round start times (TD_RAMP=1: delay = stagger * blockIdx / 512) + phase traces
cd $GRAFT_REPO_ROOT/tools
run() { timeout 60 ./conv_bench.out "$@" 2>&1 | grep -v "check vs"; }
for shape in "64 64 64 192 192 9 0 96 1 3 2" "64 64 64 192 192 9 0 96 1 3 1" "64 32 32 384 384 9 0 128 1 3 2"; do
  for s in 0 20000 40000 60000 80000; do TD_RAMP=1 run $shape $s 0 1; done
done
echo "== traces: k3 residual, ramp 0 / 50000"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 2 0 0 1
TD_RAMP=1 timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 2 50000 0 1
echo "== traces: k3 emb-silu epilogue"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 1 0 0 0
echo "== traces: k3 plain epilogue (stores only)"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 0 0 0 0
echo "== traces: k6 32x32 bn128"
timeout 60 ./cb_trace.out 64 32 32 384 384 9 0 128 1 3 2 0 0 1
echo "== traces: big tile k6 384 64x64"
timeout 60 ./cb_trace.out 64 64 64 384 384 9 0 128 1 2 1 0 0 0
