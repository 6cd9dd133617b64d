#!/bin/bash
# Base U-Net forward at batch 1 ... 64 (BASELINE configs[1] ... the bench batch): wall time per forward inside the captured 20-step sampler graph,
# kernel time per forward and conv flavours (engine profile mode), HBM bytes per forward (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
# FETCH doubled: gfx950 correction of the MI355X guide).  Runs ON THE GPU BOX; writes gpurun_out/batch_sweep.txt.   usage: tools/batch_sweep.sh [batches...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/batch_sweep; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BATCHES=${@:-1 2 4 8 16 32 64}
for n in $BATCHES; do
  B="python $R/bench.py --workload tiles --tiles-per-step $n --no-cpu-baseline --no-kernel-profile --no-latency"
  timeout 300 $B --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_$n.json
  TD_TOP=200 timeout 300 python $R/tools/profile_ops.py $n bf16 2>/dev/null | grep -v amdgpu.ids > $OUT/perop_$n.txt
  for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${n}_$c -- $B --steps 1 --warmup 1 > $OUT/pmc_${n}_$c.log 2>&1; done
done
python3 - $BATCHES <<'PY'
import csv, glob, json, os, re, sys, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); out = R + "/gpurun_out/batch_sweep"
GF = 193.654   # algorithmic GFLOP of one base forward on one tile
rows = []
for n in map(int, sys.argv[1:]):
    b = json.loads(open(f"{out}/bench_{n}.json").read())
    ms_fwd = b["ms_per_step"] / 20
    per = open(f"{out}/perop_{n}.txt").read().splitlines()
    kt = float(re.search(r"([0-9.]+) ms kernel time", per[0]).group(1))
    fl = collections.Counter(re.search(r" (f\d\w*) bn", l).group(1) + ("+splitK" if int(re.search(r" ks(\d+) ", l).group(1)) > 1 else "") for l in per[1:] if " [" in l)
    tr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = 0.0
        for f in glob.glob(f"{out}/pmc_{n}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if ("td::" in r["Kernel_Name"] or "_ZN2td" in r["Kernel_Name"]) and r["Counter_Name"] == c: v += float(r["Counter_Value"])
        tr[c] = v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 40 / 1e6   # 2 bench steps x 20 forwards; KB -> MB
    rows.append((n, ms_fwd, kt, n * GF / ms_fwd, tr["FETCH_SIZE"], tr["WRITE_SIZE"], dict(fl)))
with open(R + "/gpurun_out/batch_sweep.txt", "w") as o:
    o.write("base U-Net (30m config) forward vs batch, bf16, MI355X.  wall = ms per forward inside the captured 20-step sampler graph (bench.py --workload tiles);\n"
            "kernel = sum of per-launch HIP-event times in eager profile mode; TF/s = batch x 193.654 GFLOP / wall; HBM MB per forward = rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE\n"
            "(weights alone: 507 MB); plan = conv launches by flavour (f5c16 = conv_s16 64 px x 16 couts, f4 = small-batch conv_sb m<px/32>n<cout/32>, f2 = conv_glds big/small tile, f0 = per-tap)\n\n")
    o.write(f"{'batch':>5} {'wall ms':>9} {'kernel ms':>10} {'TF/s':>8} {'read MB':>9} {'write MB':>9}  plan\n")
    for n, w, k, tf, rd, wr, fl in rows:
        o.write(f"{n:>5} {w:>9.3f} {k:>10.3f} {tf:>8.1f} {rd:>9.0f} {wr:>9.0f}  {' '.join(f'{a}:{b}' for a, b in sorted(fl.items()))}\n")
print(open(R + "/gpurun_out/batch_sweep.txt").read())
PY
rm -rf $OUT/pmc_*/
