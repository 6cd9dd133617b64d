#!/bin/bash
# round-6 end-of-round collection in ONE gpurun call: rocprofv3 kernel trace + PMC passes first (one sampler lane, stamped with the library build id), their
# summary copied to profiles/r06_hbm_traffic_and_mfma_util.json ON THE BOX so that the bench lines taken afterwards carry `roofline.traffic`, then the
# validation run (GPU tests, smoke, bench lines incl. one fp32 line and the one-lane line the trace reproduces), the batch sweep and two option A/Bs.
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cp gpurun_out/profiles_new/hbm_traffic_and_mfma_util.json profiles/r06_hbm_traffic_and_mfma_util.json
bash tools/final_validate.sh > gpurun_out/final_validate.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-latency --engine-opts dual_stream=0 > gpurun_out/final_bench_grid8_single_lane.json 2> gpurun_out/final_bench_grid8_single_lane.err
timeout 600 python bench.py --dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/final_bench_grid8_fp32.json 2> gpurun_out/final_bench_grid8_fp32.err
timeout 900 python bench.py --workload grid32 --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-kernel-profile > gpurun_out/final_bench_grid32_n1.json 2> gpurun_out/final_bench_grid32_n1.err
timeout 900 bash tools/batch_sweep.sh > gpurun_out/batch_sweep.log 2>&1
AB_ROUNDS=2 tools/ab.sh bench -- "glds_wide=0,dual_stream=0" "glds_wide=0" "dual_stream=0" "" "glds_wide_min_wgs=1024" > gpurun_out/final_ab_wide_dual.txt 2>&1
AB_ROUNDS=2 tools/ab.sh bench --workload cascade -- "glds_wide=0" "fewcout=0" "" > gpurun_out/final_ab_cascade.txt 2>&1
TD_TOP=80 timeout 200 python tools/profile_model.py decoder 4 512 > gpurun_out/final_decoder_forward_batch4.txt 2>/dev/null
if [ -z "$SKIP_TESTS" ]; then bash tools/attn_profile.sh > gpurun_out/attn_profile.log 2>&1; fi
tail -12 gpurun_out/final_validate.log | cut -c1-600
cut -c1-300 gpurun_out/final_bench_grid8_fp32.json
tail -12 gpurun_out/batch_sweep.txt
cat gpurun_out/final_ab_wide_dual.txt gpurun_out/final_ab_cascade.txt
