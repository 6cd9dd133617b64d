#!/bin/bash
# batch 1..8: weight prefetch on a side stream (engine option weight_prefetch) A/B, graph-replayed sampler
cd $GRAFT_REPO_ROOT
for r in 1 2; do for o in 0 1; do
  echo -n "[weight_prefetch=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts weight_prefetch=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per tile x 20 steps')"
done; done
for n in 2 4 8; do for o in 0 1; do
  echo -n "[tiles/step $n weight_prefetch=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step $n --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts weight_prefetch=$o,weight_prefetch_max_px=65536 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per step')"
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x 2>&1 | tail -3
