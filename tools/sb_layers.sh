#!/bin/bash
# Battery of base-model layer shapes for tools/sb_bench.out (correctness vs the per-tap flavour, hot / cold timing, conv_glds split-K beside it).
# usage: tools/sb_layers.sh [order]     args of sb_bench: N H W Cin Cout mt nt Cin1x1 epi xform order resample glds_ks out2
B=tools/sb_bench.out; O=${1:-0}   # sb_bench.out: hipcc ... -DSB_WITH_GLDS (conv_glds split-K beside every line)
run() { echo "--- $*"; timeout 120 $B "$@" || echo "FAILED($?) $*"; }
echo "== level A 64x64"
run 1 64 64 64 192 2 2 0 0 0 $O 0 1
run 1 64 64 192 192 2 2 0 1 2 $O 0 3
run 1 64 64 192 192 2 2 0 2 0 $O 0 3 1
run 1 64 64 576 192 2 2 0 1 0 $O 0 8
run 1 64 64 192 192 2 2 576 2 0 $O 0 8
run 1 64 64 384 384 2 2 0 1 1 $O 2 4
run 1 64 64 384 384 2 2 0 2 0 $O 0 4
echo "== level B 32x32"
run 1 32 32 384 384 2 1 0 1 2 $O 0 6
run 1 32 32 384 384 2 2 0 1 2 $O 0 6
run 1 32 32 768 384 2 1 0 1 0 $O 0 12
run 1 32 32 384 384 2 1 768 2 0 $O 0 16
run 1 32 32 576 576 2 1 0 1 1 $O 2 9
echo "== level C 16x16"
run 1 16 16 576 576 1 1 0 1 2 $O 0 9
run 1 16 16 576 576 2 1 0 1 2 $O 0 9
run 1 16 16 1152 576 1 1 0 1 0 $O 0 18
run 1 16 16 576 576 1 1 1152 2 0 $O 0 27
echo "== level C 16x16, the 64 px x 16 cout flavour (conv_s16.hip: mt 0)"
run 1 16 16 576 576 0 0 0 1 2 $O 0 9
run 1 16 16 576 576 0 0 0 2 0 $O 0 9 1
run 1 16 16 1152 576 0 0 0 1 0 $O 0 18
run 1 16 16 576 576 0 0 1152 2 0 $O 0 27
run 1 16 16 384 576 0 0 0 1 0 $O 2 6
echo "== level D 8x8"
run 1 8 8 768 768 2 1 0 1 2 $O 0 12
run 1 8 8 768 768 1 1 0 1 2 $O 0 12
run 1 8 8 1536 768 2 1 0 1 0 $O 0 24
run 1 8 8 768 768 2 1 1536 2 0 $O 0 32
run 1 8 8 0 2304 2 1 768 0 0 $O 0 12
run 1 8 8 0 768 2 1 768 2 0 $O 0 12
run 1 8 8 768 768 0 0 0 1 2 $O 0 12
run 1 8 8 768 768 0 0 1536 2 0 $O 0 32
echo "== batches"
run 4 32 32 384 384 2 2 0 1 2 $O 0 2
run 4 16 16 576 576 2 1 0 1 2 $O 0 4
run 4 8 8 768 768 2 1 0 1 2 $O 0 12
run 16 16 16 576 576 2 2 0 1 2 $O 0 1
run 16 8 8 768 768 2 2 0 1 2 $O 0 4
run 3 40 40 192 192 2 2 0 2 0 $O 0 0 1
run 2 24 24 128 320 1 1 64 2 0 $O 0 0
run 2 24 24 128 320 0 0 64 2 0 $O 0 0
run 3 9 9 256 192 0 0 0 2 0 $O 0 0 1
