#!/bin/bash
# round 6, experiment 12: the persistent tile loop inside the engine -- the new bit-identity test and the tests that cover the files it touched, the decoder
# model's per-layer table with the loop off / on, the cascade bench off / on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp12.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py -x -q -m gpu -k "persistent or wide or batch64 or decoder_window or config2" -s > gpurun_out/r06_exp12_tests.txt 2>&1
grep -E "persistent loop|wide tile|passed|failed|Error|error|assert" gpurun_out/r06_exp12_tests.txt | head -30 >> $O
for o in "" "glds_wide_persist=1"; do echo "[$o]" >> $O; TD_OPTS=$o TD_TOP=80 timeout 200 python tools/profile_model.py decoder 4 512 2>/dev/null > gpurun_out/r06_exp12_decoder_b4_${o:-default}.txt; head -1 gpurun_out/r06_exp12_decoder_b4_${o:-default}.txt >> $O; done
for o in "" "glds_wide_persist=1"; do echo "[$o]" >> $O; TD_OPTS=$o TD_TOP=80 timeout 200 python tools/profile_model.py decoder 8 512 2>/dev/null | head -1 >> $O; done
AB_ROUNDS=2 tools/ab.sh bench --workload cascade -- "" "glds_wide_persist=1" >> $O 2>&1
cat $O
