#!/bin/bash
# batch-64 forward kernel time (eager, HIP events around every launch) under plan options, interleaved twice
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for o in "" "producer_act=0" "pp=1" "pp=2"; do
  echo -n "[$o] "; TD_OPTS="$o" timeout 120 python tools/profile_ops.py 64 bf16 2>/dev/null | head -1
done
done
