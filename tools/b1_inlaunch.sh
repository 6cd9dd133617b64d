#!/bin/bash
# batch-1 latency: in-launch split-K reduction (engine option splitk_inlaunch) A/B, interleaved; + per-op kernel time
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "base_forward or tile_steps or layer or unet" 2>&1 | tail -3
for r in 1 2; do for o in 0 1; do
  echo -n "[splitk_inlaunch=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts splitk_inlaunch=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per tile x 20 steps')"
done; done
TD_OPTS="splitk_inlaunch=1" TD_TOP=12 timeout 120 python tools/profile_ops.py 1 bf16 2>/dev/null
TD_OPTS="splitk_inlaunch=0" TD_TOP=3 timeout 120 python tools/profile_ops.py 1 bf16 2>/dev/null
