#!/bin/bash
# round 3: phase stagger of the two co-resident workgroups of a CU (arg 12 = cycles) and backwards walk of alternate layers (arg 13 = 2)
# args: N H W Cin Cout taps xform bn ksplit flavor epi stagger chain out2
cd $GRAFT_REPO_ROOT/tools
run() { timeout 60 ./conv_bench.out "$@" 2>&1 | grep -v "check vs"; }
echo "== stagger sweep (flavor 3 = 4-wave tile, two workgroups per CU)"
for shape in "64 64 64 192 192 9 0 96 1 3 2" "64 64 64 192 192 9 0 96 1 3 1" "64 64 64 384 192 9 0 96 1 3 1" "64 32 32 384 384 9 0 128 1 3 2" "64 32 32 384 384 9 0 128 1 3 1" "64 16 16 576 576 9 0 96 1 3 1" "64 16 16 576 576 9 0 96 1 3 2"; do
  for s in 0 10000 20000 30000 40000 60000 0; do run $shape $s 0 1; done
done
echo "== 384->384 at 64x64, big vs small tile, stagger on the small one"
run 64 64 64 384 384 9 0 128 1 2 1 0 0 0
for s in 0 20000 40000 60000; do run 64 64 64 384 384 9 0 128 1 3 1 $s 0 0; done
echo "== chain x->y->x: same walk order (1) vs second layer backwards (2); 0 = same input every launch"
for shape in "64 64 64 192 192 9 0 96 1 3 2" "64 64 64 192 192 9 0 96 1 2 2" "64 64 64 384 384 9 0 128 1 2 1" "64 32 32 384 384 9 0 128 1 3 2"; do
  for c in 0 1 2 1 2; do run $shape 0 $c 1; done
done
echo "== chain + stagger"
for c in 1 2; do run 64 64 64 192 192 9 0 96 1 3 2 30000 $c 1; done
echo "== phase trace, k3 residual layer, stagger 0 / 30000"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 2 0 0 1
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 2 30000 0 1
