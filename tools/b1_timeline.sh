#!/bin/bash
# rocprofv3 kernel trace of the single-tile sampler (one 64x64 tile x 20 steps in one captured graph): per-kernel durations and the idle gaps between
# consecutive kernels of the LAST replay.  Runs ON THE GPU BOX; writes gpurun_out/b1_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/b1_tl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --workload tiles --tiles-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency ${1:+--engine-opts $1} > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); out = R + "/gpurun_out/b1_tl"
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# last sampler replay = the last 20 forwards: find the last 1700 kernels or so; take the tail starting at the last big gap (> 200 us)
segs, cur = [], [rows[0]]
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 200_000: segs.append(cur); cur = []
    cur.append(rows[i])
segs.append(cur)
tail = [s for s in segs if len(s) >= 1000][-1]   # the last whole sampler replay (~1.7 k kernels)
span = (tail[-1][1] - tail[0][0]) / 1e3
busy = sum(e - s for s, e, _ in tail) / 1e3
gaps = [(tail[i][0] - tail[i - 1][1]) / 1e3 for i in range(1, len(tail))]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in tail: a = agg[k]; a[0] += 1; a[1] += (e - s) / 1e3
with open(R + "/gpurun_out/b1_timeline.txt", "w") as o:
    o.write(f"last graph replay: {len(tail)} kernels, span {span:.0f} us, kernel time {busy:.0f} us ({100 * busy / span:.1f} %), idle between kernels {sum(gaps):.0f} us (mean gap {sum(gaps) / len(gaps):.2f} us, median {sorted(gaps)[len(gaps) // 2]:.2f})\n")
    o.write("kernel,calls,total_us,avg_us\n")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): o.write(f"{k[:110]},{n},{t:.1f},{t / n:.2f}\n")
    o.write("gap histogram (us): " + " ".join(f"<{b}:{sum(1 for g in gaps if a <= g < b)}" for a, b in ((0, 1), (1, 1.5), (1.5, 2), (2, 3), (3, 5), (5, 10), (10, 1e9))) + "\n")
print(open(R + "/gpurun_out/b1_timeline.txt").read())
PY
rm -rf $OUT/kt
