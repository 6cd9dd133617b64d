#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x 2>&1 | tail -3
for r in 1 2; do for o in 0 1; do
  echo -n "[glds_tiny=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts glds_tiny=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per tile x 20 steps')"
done; done
for n in 2 4 8; do for o in 0 1; do echo -n "[batch $n glds_tiny=$o] "; TD_OPTS="glds_tiny=$o" TD_TOP=0 timeout 120 python tools/profile_ops.py $n bf16 2>/dev/null | head -1; done; done
TD_OPTS="glds_tiny=1" TD_TOP=90 timeout 120 python tools/profile_ops.py 1 bf16 2>/dev/null > gpurun_out/per_op_batch1_r03.txt; head -8 gpurun_out/per_op_batch1_r03.txt
