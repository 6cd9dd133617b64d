#!/bin/bash
# round 5, experiment 8: planner thresholds of the small-grid flavours at the cascade's latent-stage batch sizes (engine options only, same library):
# kernel time per base forward (tools/profile_ops.py) at batch 4 / 8 / 16 / 32 under option sets.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp8.txt; : > $O
for n in 4 8 16 32; do
  for o in "" "sb_m4=0" "s16=0" "sb_target_wgs=96" "sb_target_wgs=256" "s16_min_wgs=96" "sb=0"; do
    echo "[batch $n | ${o:-default}] $(TD_OPTS=$o TD_TOP=1 timeout 200 python tools/profile_ops.py $n bf16 2>/dev/null | head -1)" >> $O
  done
done
cat $O
