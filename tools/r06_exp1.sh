#!/bin/bash
# round 6, experiment 1: (A) configs[3] in the DEFAULT plan: one-rank canvas vs 2 / 4 simulated ranks bit for bit (review item 2), the advisor-fix tests;
# (B) dual_stream A/B on grid8 in both orders + what it does to the bits (review item 4); (C) the strong-scaling anchor under four option sets.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp1.txt; : > $O
echo "# (A) tests" >> $O
timeout 1500 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "config3 or dry_run" -s > gpurun_out/r06_exp1_tests.txt 2>&1
grep -E "configs\[3\]|passed|failed|error" gpurun_out/r06_exp1_tests.txt >> $O
timeout 900 python -m pytest tests/test_world_pipeline_gpu.py tests/test_gpu_small_batch.py -x -q -m gpu > gpurun_out/r06_exp1_tests2.txt 2>&1
tail -3 gpurun_out/r06_exp1_tests2.txt >> $O
echo "# (B) dual_stream A/B, grid8 (python bench.py), order A B A B then B A B A" >> $O
AB_ROUNDS=2 tools/ab.sh bench -- "" "dual_stream=1" >> $O 2>&1
AB_ROUNDS=2 tools/ab.sh bench -- "dual_stream=1" "" >> $O 2>&1
echo "# (B2) bits: grid8 canvas with dual_stream 0 / 1 (batch 64 plan vs two batch-32 lanes)" >> $O
timeout 600 python - >> $O 2>&1 <<'PY'
import torch, terrain_diffusion_amd as td
from terrain_diffusion_amd.engine import get_engine
from terrain_diffusion_amd.synthetic import synthetic_state_dict, synthetic_cond_grid
from oracle.unet import BASE_CONFIG
eng = get_engine("cuda:0")
m = td.EDMUnet2D(**dict(BASE_CONFIG), dtype="bf16", device="cuda:0"); m.load_state_dict(synthetic_state_dict(m, seed=1234))
sch = td.EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
cond = synthetic_cond_grid(8, 8, device="cuda:0")
kw = dict(cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5), steps=20, tile_size=64)
ys = {}
for d in (0, 1, 0):
    eng.set_option("dual_stream", d)
    y = td.sample_base_diffusion(m, sch, (1, 5, 288, 288), cond, noise_seed=42 + 5819, **kw)
    if d in ys: print("dual_stream", d, "repeat equal:", bool(torch.equal(ys[d], y)))
    ys[d] = y
a, b = ys[0], ys[1]
print("dual_stream 1 vs 0: differing values", int((a != b).sum()), "of", a.numel(), " rel-RMS", float(((a - b).pow(2).mean() / a.pow(2).mean()).sqrt()))
PY
echo "# (C) strong-scaling anchor workload (grid32 on one rank, 1 step) under option sets" >> $O
for o in "batch_invariant=1,dual_stream=1" "batch_invariant=0,dual_stream=1" "batch_invariant=0,dual_stream=0" "batch_invariant=1,dual_stream=0"; do
  echo -n "[$o] " >> $O
  timeout 600 python bench.py --workload grid32 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --engine-opts "$o" 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" >> $O
done
cat $O
