#!/bin/bash
# split-K slice boundaries by K-steps instead of K-groups for launches that mix 3x3 and 1x1 groups (engine option splitk_weighted)
cd $GRAFT_REPO_ROOT
for o in 0 1 0 1; do
  echo -n "[1 tile x 20 steps, splitk_weighted=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts splitk_weighted=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms')"
done
for o in 0 1; do echo -n "[grid8 splitk_weighted=$o] "; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --engine-opts splitk_weighted=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step frac', d['roofline']['frac'])"; done
for n in 1 4; do for o in 0 1; do echo -n "[batch $n splitk_weighted=$o] "; TD_OPTS="splitk_weighted=$o" TD_TOP=0 timeout 120 python tools/profile_ops.py $n bf16 2>/dev/null | head -1; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_bench_config.py -q -x 2>&1 | tail -3
