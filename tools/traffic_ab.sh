#!/bin/bash
# HBM-side traffic of the conv_glds family under engine option sets (separate rocprofv3 --pmc passes, FETCH_SIZE x 2 per the gfx950 correction,
# KB -> bytes), per launch, beside the bench value:   tools/traffic_ab.sh "" "producer_act=0" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/traffic_ab; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for o in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/p${i}_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency ${o:+--engine-opts "$o"} > $O/p${i}_$c.log 2>&1
  done
  timeout 600 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency ${o:+--engine-opts "$o"} 2>/dev/null | tail -1 > $O/b$i.json
  i=$((i+1))
done
python3 - "$@" <<'PY'
import csv, glob, json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); O = R + "/gpurun_out/traffic_ab"
for i, o in enumerate(sys.argv[1:]):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        n = 0; v = 0.0
        for f in glob.glob(f"{O}/p{i}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_glds" in r["Kernel_Name"] and r["Counter_Name"] == c: n += 1; v += float(r["Counter_Value"])
        tot[c] = (v * 1024 * (2 if c == "FETCH_SIZE" else 1) / max(1, n), n)
    d = json.loads(open(f"{O}/b{i}.json").read())
    print(f"[{o or 'defaults'}] {d['value']:.2f} MP/s; conv_glds launches counted {tot['FETCH_SIZE'][1]}: read {tot['FETCH_SIZE'][0] / 1e6:.1f} MB + written {tot['WRITE_SIZE'][0] / 1e6:.1f} MB = {(tot['FETCH_SIZE'][0] + tot['WRITE_SIZE'][0]) / 1e6:.1f} MB per launch")
PY
rm -rf $O/p*
