#!/bin/bash
# round 6, experiment 13: the few-cout flavour (decoder model's output conv): tests, decoder per-layer table, cascade bench with it off / on
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp13.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py tests/test_world_pipeline_gpu.py -x -q -m gpu -k "fewcout or persistent or decoder or cascade or world or stage" -s > gpurun_out/r06_exp13_tests.txt 2>&1
grep -E "few-cout|persistent loop|passed|failed|Error|error|assert" gpurun_out/r06_exp13_tests.txt | head -30 >> $O
for o in "fewcout=0" ""; do echo "[$o]" >> $O; TD_OPTS=$o TD_TOP=80 timeout 200 python tools/profile_model.py decoder 4 512 2>/dev/null > gpurun_out/r06_exp13_decoder_b4_${o:-default}.txt; head -1 gpurun_out/r06_exp13_decoder_b4_${o:-default}.txt >> $O; grep out_conv gpurun_out/r06_exp13_decoder_b4_${o:-default}.txt >> $O; done
AB_ROUNDS=2 tools/ab.sh bench --workload cascade -- "fewcout=0" "" >> $O 2>&1
cat $O
