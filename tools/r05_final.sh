#!/bin/bash
# round-5 end-of-round collection in ONE gpurun call: rocprofv3 kernel trace + PMC passes first (stamped with the library build id), their summary copied
# to profiles/r05_hbm_traffic_and_mfma_util.json ON THE BOX so that the bench lines taken afterwards carry `roofline.traffic`, then the validation run
# (GPU tests, smoke, bench lines incl. one fp32 line), then the batch sweep (all seven rows from this one build).  Everything lands in gpurun_out/.
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cp gpurun_out/profiles_new/hbm_traffic_and_mfma_util.json profiles/r05_hbm_traffic_and_mfma_util.json
bash tools/final_validate.sh > gpurun_out/final_validate.log 2>&1
timeout 600 python bench.py --dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/final_bench_grid8_fp32.json 2> gpurun_out/final_bench_grid8_fp32.err
timeout 900 bash tools/batch_sweep.sh > gpurun_out/batch_sweep.log 2>&1
tail -12 gpurun_out/final_validate.log | cut -c1-600
cut -c1-300 gpurun_out/final_bench_grid8_fp32.json
tail -12 gpurun_out/batch_sweep.txt
