#!/bin/bash
# LDS-DMA streaming of 1x1 K-segments in conv_glds (engine option glds_dma1x1): bit-identity of the network output, per-op times, bench line
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dma1x1_ab.py 2>&1 | tail -3
for o in 0 1; do echo "== glds_dma1x1=$o"; TD_OPTS="glds_dma1x1=$o" TD_TOP=90 timeout 200 python tools/profile_ops.py 64 bf16 2>/dev/null > gpurun_out/per_op_b64_dma$o.txt; head -1 gpurun_out/per_op_b64_dma$o.txt; grep "dec.*conv_res1" gpurun_out/per_op_b64_dma$o.txt | head -${ROWS:-16}; done
for o in 0 1 0 1; do echo -n "[bench glds_dma1x1=$o] "; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --engine-opts glds_dma1x1=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step frac', d['roofline']['frac'])"; done
