#!/bin/bash
# one workgroup per CU (extra LDS) vs two: is the tap loop of a lone workgroup MFMA-bound?
cd $GRAFT_REPO_ROOT/tools
for x in 0 40000; do
echo "== extra LDS $x"
TD_EXTRA_LDS=$x timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 2 0 0 1
TD_EXTRA_LDS=$x timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 3 0 0 0 0
TD_EXTRA_LDS=$x timeout 60 ./cb_trace.out 64 32 32 384 384 9 0 128 1 3 1 0 0 0
done
echo "== big tiles (always one per CU)"
timeout 60 ./cb_trace.out 64 64 64 192 192 9 0 96 1 2 2 0 0 1
timeout 60 ./cb_trace.out 64 32 32 384 384 9 0 128 1 2 1 0 0 0
