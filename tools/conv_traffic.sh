#!/bin/bash
# HBM-side bytes of single conv layers (tools/conv_bench192.out = conv_bench.hip built with -DTD_BN192) against their algorithmic bytes:
# FETCH_SIZE x 2 (calibrated: tools/pmc_calib.sh) and WRITE_SIZE per launch, separate rocprofv3 passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$R/tools/conv_bench192.out; O=$R/gpurun_out/conv_traffic; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
while read -r args; do
  [ -z "$args" ] && continue
  for c in FETCH_SIZE WRITE_SIZE; do timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/l${i}_$c -- $B $args > $O/l${i}_$c.log 2>&1; done
  echo "$args" > $O/l$i.args
  i=$((i+1))
done <<'LAYERS'
64 64 64 192 192 9 0 96 1 3 1
64 64 64 192 192 9 0 96 1 2 1
64 64 64 192 192 9 0 64 1 3 1
64 64 64 192 192 1 0 96 1 3 1
64 64 64 384 192 9 0 96 1 3 1
64 32 32 384 384 9 0 128 1 3 1
64 32 32 384 384 9 0 96 1 3 1
64 16 16 576 576 9 0 96 1 3 1
LAYERS
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); O = R + "/gpurun_out/conv_traffic"
print("N HxW Cin->Cout taps bn tile | algorithmic: in + weights + out MB | measured: read (FETCH_SIZE x 2) + written MB per launch | read / (in + weights), written / out")
for i in range(32):
    if not os.path.exists(f"{O}/l{i}.args"): break
    a = open(f"{O}/l{i}.args").read().split(); N, H, W, Ci, Co, taps, xf, bn, ks, fl = (int(x) for x in a[:10])
    got = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        n = 0; v = 0.0
        for f in glob.glob(f"{O}/l{i}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_glds" in r["Kernel_Name"] and r["Counter_Name"] == c: n += 1; v += float(r["Counter_Value"])
        got[c] = v * 1024 * (2 if c == "FETCH_SIZE" else 1) / max(1, n) / 1e6
    inp = N * H * W * Ci * 2 / 1e6; wt = Ci * taps * Co * 2 / 1e6; out = N * H * W * Co * 2 / 1e6
    print(f"{N} {H}x{W} {Ci}->{Co} taps{taps} bn{bn} {'big 256px' if fl == 2 else 'small 128px'} | {inp:.1f} + {wt:.2f} + {out:.1f} | {got['FETCH_SIZE']:.1f} + {got['WRITE_SIZE']:.1f} | {got['FETCH_SIZE'] / (inp + wt):.2f}x, {got['WRITE_SIZE'] / out:.2f}x")
PY
rm -rf $O/l*_FETCH_SIZE $O/l*_WRITE_SIZE
