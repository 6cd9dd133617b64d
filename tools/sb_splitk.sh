#!/bin/bash
# deep levels at batch 1: in-workgroup split only vs split-K over workgroups on top (sb_ks) -- tools/sb_fast.out = sb_bench.hip without conv_glds
B=tools/sb_fast.out
run() { timeout 120 $B "$@" | grep -v "sumsq\|out2" || echo "FAILED($?) $*"; }
echo "== level D 8x8 768->768 (K 12 groups)"; for ks in 1 3 6; do for cfg in "2 1" "1 1"; do run 1 8 8 768 768 $cfg 0 1 2 0 0 0 0 $ks; done; done
echo "== level D dec 1536->768 (24 groups)"; for ks in 1 4 8 12; do run 1 8 8 1536 768 2 1 0 1 0 0 0 0 0 $ks; done
echo "== level D res1 768 + 1536 1x1"; for ks in 1 4 8; do run 1 8 8 768 768 2 1 1536 2 0 0 0 0 0 $ks; done
echo "== level C 16x16 576->576 (9 groups)"; for ks in 1 3; do for cfg in "2 1" "1 1"; do run 1 16 16 576 576 $cfg 0 1 2 0 0 0 0 $ks; done; done
echo "== level C dec 1152->576 (18 groups)"; for ks in 1 3 6; do for cfg in "2 1" "1 1"; do run 1 16 16 1152 576 $cfg 0 1 0 0 0 0 0 $ks; done; done
echo "== level B 32x32 768->384 (12 groups)"; for ks in 1 2 3; do run 1 32 32 768 384 2 1 0 1 0 0 0 0 0 $ks; done
echo "== level B 32x32 768->384 m2n2"; for ks in 1 2 4; do run 1 32 32 768 384 2 2 0 1 0 0 0 0 0 $ks; done
