#!/bin/bash
# round-4 end-of-round collection in ONE gpurun call: rocprofv3 kernel trace + PMC passes first (stamped with the library build id), their
# summary copied to profiles/r04_hbm_traffic_and_mfma_util.json ON THE BOX so that the bench lines taken afterwards carry `roofline.traffic`,
# then the validation run (GPU tests, smoke, bench lines).  Everything lands in gpurun_out/.
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cp gpurun_out/profiles_new/hbm_traffic_and_mfma_util.json profiles/r04_hbm_traffic_and_mfma_util.json
bash tools/final_validate.sh > gpurun_out/final_validate.log 2>&1
tail -12 gpurun_out/final_validate.log | cut -c1-600
