#!/bin/bash
# A/B on one box: two concurrent half-batch lanes (engine option dual_stream, default 1) vs one lane
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for d in 0 1; do
  echo -n "grid8 dual_stream=$d: "; python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts dual_stream=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
for d in 0 1; do
  echo -n "tiles dual_stream=$d: "; python bench.py --workload tiles --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts dual_stream=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
