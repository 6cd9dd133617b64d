#!/bin/bash
# round 3: walk order alternated from conv to conv (engine option walk_alternate) -- per-forward kernel time and the default bench, interleaved A/B
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for o in "walk_alternate=0" "walk_alternate=1"; do
  echo -n "[$o] "; TD_OPTS="$o" timeout 120 python tools/profile_ops.py 64 bf16 2>/dev/null | head -1
done
done
for r in 1 2; do
for o in "walk_alternate=0" "walk_alternate=1"; do
  echo -n "[$o] "; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --engine-opts $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_launch_us'))"
done
done
TD_OPTS="walk_alternate=1" TD_TOP=90 timeout 120 python tools/profile_ops.py 64 bf16 2>/dev/null > gpurun_out/per_op_batch64_r03.txt
