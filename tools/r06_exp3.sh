#!/bin/bash
# round 6, experiment 3: the whole GPU suite under the new defaults (two sampler lanes, wide tile), then the wide tile A/B again in both lane modes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp3.txt; : > $O
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_exp3_tests.txt 2>&1
tail -15 gpurun_out/r06_exp3_tests.txt >> $O
echo "# bench A/B single lane, wide tile off / on" >> $O
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide=0,dual_stream=0" "dual_stream=0" >> $O 2>&1
echo "# bench A/B two lanes, wide tile off / on / also at the 16x16 level (min 384 workgroups)" >> $O
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide=0" "" "glds_wide_min_wgs=384" >> $O 2>&1
echo "# cascade, wide tile off / on" >> $O
AB_ROUNDS=1 tools/ab.sh bench --workload cascade -- "glds_wide=0" "" >> $O 2>&1
cat $O
