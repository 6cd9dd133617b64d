#!/bin/bash
# ablation of the conv_glds epilogue: arithmetic / stores / operand loads removed one at a time (tools/conv_bench.hip builds with -DTD_ABL_EPI_x)
cd $GRAFT_REPO_ROOT/tools
for shape in "64 64 64 384 384 9 0 128" "64 64 64 192 192 9 0 96"; do
  for epi in 1 2; do
    for b in conv_bench cbe_NOVALU cbe_NOST cbe_NOLD cbe_NOMEM cb_noepi; do
      echo -n "$b: "; timeout 60 ./$b.out $shape 1 2 $epi 2>&1 | grep -v "check vs"
    done
  done
done
