"""Per-op kernel-time breakdown of the base U-Net (profile mode: HIP events around every launch, eager)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from oracle.unet import BASE_CONFIG, synth_state_dict
from oracle import rng

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = 3
for kv in os.environ.get('TD_OPTS', '').split(','):
    if kv:
        k, v = kv.split('='); td.engine.get_engine("cuda").set_option(k, int(v))
cfg = dict(BASE_CONFIG)
m = td.EDMUnet2D(**cfg, dtype=dtype).load_state_dict(synth_state_dict(cfg, seed=1234))
eng = m.engine
x = torch.from_numpy(rng.standard_normal(7, (n, 5, 64, 64))).cuda()
c = torch.from_numpy(rng.standard_normal(8, (n, 58))).cuda()
t = torch.full((n,), 1.1)
m(x, t, [c])
eng.set_option("profile", 1)
eng.profile_read(reset=True)
for _ in range(reps):
    m(x, t, [c])
rows = eng.profile_ops()
conv_ms, conv_n, other_ms, other_n = eng.profile_read()
tot = sum(r[1] for r in rows)
print(f"batch {n} {dtype}: {tot / reps:.3f} ms kernel time per forward ({conv_n // reps} conv launches, {other_n // reps} other)")
order = {}
import re
for r in sorted(rows, key=lambda r: -r[1])[:int(os.environ.get("TD_TOP", "40"))]:
    us = r[1] / reps * 1e3
    m = re.search(r" gf([0-9.]+)\]", r[0])
    tf = f"{float(m.group(1)) / us * 1e3:7.1f} TF/s" if m and us > 0 else " " * 12   # GFLOP / us = PFLOP/s
    print(f"{us:9.1f} us {tf}  {r[0]}")
