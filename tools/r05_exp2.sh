#!/bin/bash
# round 5, experiment 2 (one gpurun call): (A) conv_glds after the LDS-budget fix of the DMA instantiation (bn 128: the modulation rows cost it its second
# workgroup per CU in experiment 1) -- bits and time against the round-4 kernel; (B) the deep-level latency flavour (conv_s16.hip) against the per-tap
# flavour (correctness) and against conv_sb + split-K over workgroups + reduce launch (time); (C) GPU tests; (D) bench + single-tile latency; (E) per-op tables.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_exp2.txt; : > $O
ab() {
  local envp="$1"; shift
  echo "## $envp $*" >> $O
  for b in base new; do
    env $envp timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
    env $envp TD_DUMP=gpurun_out/cb_$b.bin timeout 120 tools/conv_bench_$b.out $* | head -1 | sed "s/^/  $b: /" >> $O
  done
  cmp gpurun_out/cb_base.bin gpurun_out/cb_new.bin > /dev/null && echo "  bits: identical ($(stat -c %s gpurun_out/cb_new.bin) bytes)" >> $O || echo "  bits: DIFFER" >> $O
}
echo "# (A) conv_glds: base = round-4 kernel, new = this tree" >> $O
ab TD_SEG2=384,1 64 32 32 384 384 9 0 128 1 3 2
ab TD_SEG2=768,1 64 32 32 384 384 9 0 128 1 3 0
ab X=1 64 32 32 192 384 1 0 128 1 3 0
ab X=1 64 32 32 384 384 9 0 128 1 3 1
ab X=1 64 64 64 192 192 9 0 96 1 3 1
ab X=1 64 64 64 192 192 9 0 96 1 3 2 0 0 1
ab X=1 64 8 8 768 768 9 0 96 1 3 1
ab X=1 64 8 8 768 768 9 0 128 1 2 1
ab X=1 3 20 20 192 192 9 0 96 1 3 1
rm -f gpurun_out/cb_*.bin
echo "# (B) conv_s16 (mt = 0) vs the per-tap flavour (correctness) and vs conv_sb m2n1 + split-K over workgroups (sb_ks) + reduce (time); batch 1" >> $O
sb() { echo "## $*" >> $O; timeout 120 tools/sb_bench.out $* 2>&1 | head -6 >> $O; }
#   N H W Cin Cout mt nt Cin1x1 epi xform order resample glds_ks out2 sb_ks
sb 1 8 8 768 768 0 0 0 1
sb 1 8 8 768 768 2 1 0 1 0 1 0 0 0 9
sb 1 8 8 768 768 0 0 0 2 0 1 0 0 1
sb 1 8 8 768 768 0 0 0 1 2 1
sb 1 8 8 1536 768 0 0 0 1 0 1
sb 1 8 8 1536 768 2 1 0 1 0 1 0 0 0 9
sb 1 8 8 768 768 0 0 1536 2 0 1
sb 1 8 8 768 768 2 1 1536 2 0 1 0 0 0 9
sb 1 8 8 0 768 0 0 768 2 0 1
sb 1 8 8 0 2304 0 0 768 0 0 1
sb 1 16 16 576 576 0 0 0 1 0 1
sb 1 16 16 576 576 2 1 0 1 0 1 0 0 0 3
sb 1 16 16 576 576 0 0 0 2 2 1 0 0 1
sb 1 16 16 1152 576 0 0 0 1 0 1
sb 1 16 16 576 576 0 0 1152 2 0 1
sb 1 16 16 576 576 2 1 1152 2 0 1 0 0 0 3
sb 1 16 16 384 576 0 0 0 1 0 0 2
sb 1 9 9 256 192 0 0 0 2 0 0
sb 2 18 18 192 192 0 0 64 1 0 1
sb 3 8 8 64 64 0 0 0 0 0 0
sb 2 8 8 768 768 0 0 0 1 0 1
echo "# (C) GPU tests on the rebuilt library" >> $O
timeout 1200 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_edges.py tests/test_world_pipeline_gpu.py tests/test_gpu_bench_config.py tests/test_gpu_parity.py -x -q -m gpu -k "not config4 and not config3 and not third_order" > gpurun_out/r05_exp2_tests.txt 2>&1
tail -6 gpurun_out/r05_exp2_tests.txt >> $O
echo "# (D) bench" >> $O
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_exp2_bench.json 2> gpurun_out/r05_exp2_bench.err
python - >> $O <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_exp2_bench.json"))
    r = d["roofline"]
    print({k: d.get(k) for k in ("value", "ms_per_step", "latency_single_tile_ms")}, {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "kernel_ms_per_step", "all_conv_kernels_ms_per_step")}, d.get("roofline_single_tile", {}).get("frac"))
except Exception as e:
    print("bench parse failed", e)
PY
echo "# (E) per-op, batch 1 (s16 on / off) and batch 64" >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 1 bf16 > gpurun_out/r05_exp2_per_op_batch1.txt 2>/dev/null; head -1 gpurun_out/r05_exp2_per_op_batch1.txt >> $O
TD_OPTS=s16=0 TD_TOP=90 timeout 200 python tools/profile_ops.py 1 bf16 > gpurun_out/r05_exp2_per_op_batch1_s16off.txt 2>/dev/null; head -1 gpurun_out/r05_exp2_per_op_batch1_s16off.txt >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 2 bf16 > gpurun_out/r05_exp2_per_op_batch2.txt 2>/dev/null; head -1 gpurun_out/r05_exp2_per_op_batch2.txt >> $O
TD_OPTS=s16=0 TD_TOP=90 timeout 200 python tools/profile_ops.py 2 bf16 2>/dev/null | head -1 >> $O
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 > gpurun_out/r05_exp2_per_op_batch64.txt 2>/dev/null; head -1 gpurun_out/r05_exp2_per_op_batch64.txt >> $O
for o in "" "s16=0"; do echo "[single tile x 20 steps, $o]" >> $O; timeout 200 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts "$o" 2>/dev/null | cut -c1-230 >> $O; done
tail -120 $O
