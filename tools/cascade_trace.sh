#!/bin/bash
# rocprofv3 kernel trace of the cascade bench: GPU busy share and kernel time by kernel name
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/casc_tl; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --workload cascade --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); out = R + "/gpurun_out/casc_tl"
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
ms_step = d["ms_per_step"]
# timed region = the last 2 steps: take kernels in the last 2*ms_step window before the last kernel of the steps... approximate: last 2.2 steps of wall
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - int(2.6 * ms_step * 1e6)]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in win: a = agg[k]; a[0] += 1; a[1] += (e - s) / 1e6
busy = sum(v[1] for v in agg.values()); span = (win[-1][1] - win[0][0]) / 1e6
with open(R + "/gpurun_out/cascade_timeline.txt", "w") as o:
    o.write(f"cascade bench under rocprofv3: {d['value']} MP/s, {ms_step} ms per step; window of {span:.0f} ms: kernel time {busy:.0f} ms = {100 * busy / span:.1f} % busy, {len(win)} kernels\n")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]: o.write(f"  {t:9.2f} ms  {n:6d} calls  {t / n * 1e3:9.1f} us avg  {k}\n")
# where the GPU waits: idle gaps between consecutive kernels, by the pair of kernels around them
gaps = collections.defaultdict(lambda: [0, 0.0]); tot = {50: 0.0, 200: 0.0, 1000: 0.0}; cur_end = win[0][1]; prev = win[0][2]
for s, e, k in win[1:]:
    g = (s - cur_end) / 1e3
    if g > 50:
        a = gaps[(prev, k)]; a[0] += 1; a[1] += g / 1e3
        for th in tot:
            if g > th: tot[th] += g / 1e3
    if e > cur_end: cur_end = e; prev = k
with open(R + "/gpurun_out/cascade_timeline.txt", "a") as o:
    o.write(f"idle gaps: {tot[50]:.1f} ms in gaps > 50 us, {tot[200]:.1f} ms in gaps > 200 us, {tot[1000]:.1f} ms in gaps > 1 ms; by (kernel before -> kernel after):\n")
    for (a_, b_), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]: o.write(f"  {t:8.2f} ms  {n:5d} gaps  {a_[:48]} -> {b_[:48]}\n")
print(open(R + "/gpurun_out/cascade_timeline.txt").read())
PY
rm -rf $OUT/kt
