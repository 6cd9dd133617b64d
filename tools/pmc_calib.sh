#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/pmc_calib.out's kernels against their known byte counts
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_calib; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- $R/tools/pmc_calib.out > $O/$c.log 2>&1; done
grep "known bytes" $O/FETCH_SIZE.log
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); O = R + "/gpurun_out/pmc_calib"
B = 3 << 30; npix = B // 384
known = {"calib_read_stream": B, "calib_read_patch": npix * 128, "calib_write_stream": B, "calib_write_epilogue": npix * 384}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == c and k in known:
                v = float(r["Counter_Value"]) * 1024
                print(f"{c:10s} {k:22s} counter {v / 1e6:10.1f} MB   known {known[k] / 1e6:10.1f} MB   counter / known = {v / known[k]:.3f}")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
