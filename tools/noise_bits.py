"""How often does the device normal stream differ from the host stream (oracle C = reference arithmetic), and by how much?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import terrain_diffusion_amd as td
from oracle import rng
tot = bad = 0
for seed in (1, 42, 5861, 99991):
    n = 2_000_000
    g = td.standard_normal(seed, n).view(np.uint32) if hasattr(td, "standard_normal") else None
    from terrain_diffusion_amd.noise import standard_normal
    g = standard_normal(seed, n)
    r = rng.standard_normal(seed, (n,)).astype(np.float32)
    d = (g.view(np.int32).astype(np.int64) - r.view(np.int32).astype(np.int64))
    nb = int((d != 0).sum()); tot += n; bad += nb
    idx = np.nonzero(d)[0][:5]
    print(f"seed {seed}: {nb} of {n} differ ({nb / n:.2e}); max |ulp| {np.abs(d).max()}; first at {idx.tolist()} values {g[idx].tolist()} vs {r[idx].tolist()}")
print(f"total mismatch rate {bad / tot:.3e}")
