#!/bin/bash
# 192-cout tiles of conv_glds (build: hipcc ... -DTD_BN192 tools/conv_bench.hip -o tools/conv_bench192.out) against the planner's choices on the
# 64x64 level of the base model at batch 64.  args: N H W Cin Cout taps xform bn ksplit flavor(2 big / 3 small) epi stagger chain out2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B=tools/conv_bench192.out
run() { echo "## $*"; for r in 1 2; do timeout 120 $B "$@" | grep -v "^ *$"; done; }
echo "=== enc 64x64 192->192 3x3 (k3), emb+silu epilogue"
run 64 64 64 192 192 9 0 96 1 3 1
run 64 64 64 192 192 9 0 96 1 2 1
TD_CMP_BN=96 run 64 64 64 192 192 9 0 192 1 3 1
TD_CMP_BN=96 run 64 64 64 192 192 9 0 192 1 2 1
echo "=== enc 64x64 192->192 3x3 (k3), residual epilogue + second output, normed input"
run 64 64 64 192 192 9 2 96 1 3 2 0 0 1
TD_CMP_BN=96 run 64 64 64 192 192 9 2 192 1 3 2 0 0 1
TD_CMP_BN=96 run 64 64 64 192 192 9 2 192 1 2 2 0 0 1
echo "=== dec 64x64 384->192 3x3 (k6), emb+silu"
run 64 64 64 384 192 9 0 96 1 3 1
run 64 64 64 384 192 9 0 192 1 3 1
run 64 64 64 384 192 9 0 192 1 2 1
echo "=== dec 64x64 384->384 3x3 (k6), bn128 small vs bn192"
run 64 64 64 384 384 9 0 128 1 3 1
run 64 64 64 384 384 9 0 192 1 3 1
run 64 64 64 384 384 9 0 192 1 2 1
echo "=== 32x32 384->384 (k6) bn128 vs 192"
run 64 32 32 384 384 9 0 128 1 3 1
run 64 32 32 384 384 9 0 192 1 3 1
run 64 32 32 384 384 9 0 192 1 2 1
echo "=== 16x16 576->576 (k9) bn96 small (768 wgs) vs bn192 (384 wgs, one per CU)"
run 64 16 16 576 576 9 0 96 1 3 1
run 64 16 16 576 576 9 0 192 1 3 1
run 64 16 16 576 576 9 0 192 1 2 1
