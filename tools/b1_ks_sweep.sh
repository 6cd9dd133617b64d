#!/bin/bash
# batch-1: cap of the split-K factor vs forward kernel time (partials traffic vs parallelism)
cd $GRAFT_REPO_ROOT
for o in "glds_splitk_max=32" "glds_splitk_max=16" "glds_splitk_max=8" "glds_splitk_max=4" "glds_splitk_max=2" "glds_splitk_max=32,glds_splitk_min_groups=2" "glds_splitk_max=32,glds_splitk_min_groups=3" "glds_variant=0" "glds_splitk_max=32"; do
  echo -n "[$o] "; TD_OPTS="$o" TD_TOP=0 timeout 120 python tools/profile_ops.py 1 bf16 2>/dev/null | head -1
done
