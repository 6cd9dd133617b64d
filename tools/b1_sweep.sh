#!/bin/bash
# batch-1 forward kernel time (eager, HIP events around every launch) under plan options
cd $GRAFT_REPO_ROOT
for o in "" "glds_splitk_from_groups=2" "glds_splitk_from_groups=1" "glds_splitk_from_groups=2,glds_splitk_max=64" "glds_min_wgs=100000" "glds_min_wgs=100000,splitk_target_wgs=1024" "glds_min_wgs=100000,splitk_target_wgs=256" "splitk=0"; do
  echo -n "[$o] "; TD_OPTS="$o" timeout 120 python tools/profile_ops.py 1 bf16 2>/dev/null | head -4 | tr '\n' ';'; echo
done
