#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for f in 3 7; do
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 $f 2 0 0 1 | grep -v "check vs"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 $f 1 0 0 0 | grep -v "check vs"
timeout 60 ./cb_trace.out 64 16 16 576 576 9 0 96 1 $f 1 0 0 0 | grep -v "check vs"
done
