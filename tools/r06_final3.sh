#!/bin/bash
# round 6, the call after the few-cout flavour's staging fix: its test and the decoder model's per-layer table first, then the end-of-round collection
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_fewcout_after_unroll.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "fewcout or decoder_window" -s 2>&1 | grep -E "few-cout|passed|failed|rror" >> $O
for o in "fewcout=0" ""; do echo "[$o]" >> $O; TD_OPTS=$o TD_TOP=80 timeout 200 python tools/profile_model.py decoder 4 512 2>/dev/null | grep -E "kernel time|out_conv" >> $O; done
bash tools/r06_final.sh
echo ====; cat $O
