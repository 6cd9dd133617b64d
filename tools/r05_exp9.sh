#!/bin/bash
# round 5, experiment 9 (short check before the last collection): the modulation-vector GEMM on the fp32 matrix cores -- parity tests (fp32 / bf16 / fp16 goldens), time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp9.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_batch.py tests/test_gpu_edges.py -x -q -m gpu -k "not third_order" > gpurun_out/r05_exp9_tests.txt 2>&1; tail -4 gpurun_out/r05_exp9_tests.txt >> $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp9_kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency > $GRAFT_REPO_ROOT/gpurun_out/r05_exp9_bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 - >> $O <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/exp9_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"].split("(")[0]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if any(s in k for s in ("cvec", "emb_kernel")): print(f"{k[:60]:60s} calls {n:5d} avg {t / n:9.1f} us")
PY
rm -rf gpurun_out/exp9_kt
cut -c1-260 gpurun_out/r05_exp9_bench_under_rocprof.json >> $O
timeout 200 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency 2>/dev/null | cut -c120-230 >> $O
cat $O
