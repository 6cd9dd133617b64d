#!/bin/bash
# cross-segment patch prefetch (3x3 segment -> next 3x3 segment) in conv_glds: output hashes and bench of the new build vs the previous one
cd $GRAFT_REPO_ROOT
L=terrain_diffusion_amd/libtd_engine.so
cp $L /tmp/new.so
b() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step frac', d['roofline']['frac'], d['roofline'].get('library_build_id'))"; }
timeout 200 python tools/out_hash.py 2>/dev/null | tail -1
echo -n "[new] "; b
cp tools/prev_libtd_engine.so $L
timeout 200 python tools/out_hash.py 2>/dev/null | tail -1
echo -n "[prev] "; b
cp /tmp/new.so $L; echo -n "[new] "; b
cp tools/prev_libtd_engine.so $L; echo -n "[prev] "; b
cp /tmp/new.so $L
TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 2>/dev/null > gpurun_out/per_op_b64_xseg.txt; head -1 gpurun_out/per_op_b64_xseg.txt
timeout 600 python -m pytest tests/test_gpu_bench_config.py -q -x -k "ragged or tile_variants or batch64 or real_size" 2>&1 | tail -3
