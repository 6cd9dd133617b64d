#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for shape in "64 64 64 384 384 9 0 128 1 5 1" "64 64 64 192 192 9 0 96 1 5 2"; do
  for b in conv_bench cbp_nofetch cbp_nostage cbp_noepi cbp_noall cbp_trace conv_bench; do
    echo -n "$b: "; timeout 60 ./$b.out $shape | grep -v "check vs"
  done
done
