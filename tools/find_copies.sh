#!/bin/bash
# where do the __amd_rocclr_copyBuffer launches of a bench step come from?  kernel trace of one step, neighbours of every copy
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/copies; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); out = R + "/gpurun_out/copies"
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:60] for r in rows]
print(len(rows), "kernels; columns:", list(rows[0].keys()))
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print(len(idx), "copyBuffer launches")
ctx = collections.Counter()
for i in idx:
    ctx[(names[i - 1] if i else "", names[i + 1] if i + 1 < len(names) else "")] += 1
for k, v in ctx.most_common(12): print(v, k)
for i in idx[:6] + idx[-6:]:
    r = rows[i]
    print(i, r.get("Grid_Size"), r.get("Workgroup_Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", r.get("Stream_Id", r.get("Queue_Id")))
PY
rm -rf $OUT/kt
