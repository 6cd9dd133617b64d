#!/bin/bash
# LDS-DMA 1x1 path extended to split-K slices, narrow tiles and pure 1x1 convs: bit-identity (split-K on), batch-1 and grid8 A/B
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dma1x1_ab.py 2>&1 | tail -4
for o in 0 1 0 1; do
  echo -n "[1 tile x 20 steps, glds_dma1x1=$o] "; timeout 300 python bench.py --workload tiles --tiles-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts glds_dma1x1=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms')"
done
for o in 0 1 0 1; do echo -n "[grid8 glds_dma1x1=$o] "; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --engine-opts glds_dma1x1=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step frac', d['roofline']['frac'])"; done
for n in 64 1; do TD_OPTS="glds_dma1x1=1" TD_TOP=90 timeout 200 python tools/profile_ops.py $n bf16 2>/dev/null > gpurun_out/per_op_b${n}_dma2.txt; head -1 gpurun_out/per_op_b${n}_dma2.txt; done
timeout 600 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_attention.py -q -x -k "ragged or tile_variants or attention or batch64" 2>&1 | tail -3
