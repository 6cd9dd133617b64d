#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 256 100000 256 100000; do echo "== TD_ATTN_BIG_MIN=$m"; for i in 2 6 7; do TD_ATTN_BIG_MIN=$m REPS=50 python tools/attn_bench.py $i 2>/dev/null | grep CASE | sed 's/.*wall per call/wall/'; done; done
