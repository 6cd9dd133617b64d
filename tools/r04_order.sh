#!/bin/bash
# sb_order A/B at batch 1: wall time and HBM-side fetch bytes per forward
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/order; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for o in 0 1; do
  B="python $R/bench.py --workload tiles --tiles-per-step 1 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts sb_order=$o"
  $B --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sb_order=$o', d['ms_per_step'], 'ms per tile x 20 steps')"
  for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/p_${o}_$c -- $B --steps 1 --warmup 1 > $OUT/p_${o}_$c.log 2>&1; done
  python3 - $o <<'PY'
import csv, glob, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); o = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = 0.0
    for f in glob.glob(f"{R}/gpurun_out/order/p_{o}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("td::" in r["Kernel_Name"] or "_ZN2td" in r["Kernel_Name"]) and r["Counter_Name"] == c: v += float(r["Counter_Value"])
    print(f"  sb_order={o} {c}: {v * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 40 / 1e6:.0f} MB per forward")
PY
done
rm -rf $OUT
