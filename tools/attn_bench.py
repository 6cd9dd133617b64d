"""Attention cases for the MFMA utilisation report (VERDICT row N1).  Run plain for wall-clock per call, or under
`rocprofv3 --kernel-trace --stats` / `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (tools/attn_profile.sh) for per-kernel numbers."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terrain_diffusion_amd.attention import attention  # noqa: E402
from terrain_diffusion_amd.engine import get_engine  # noqa: E402

CASES = [("terrain 8x8 block, 64 tiles x 12 heads", 64, 12, 64, 64, 64, True), ("terrain 16x16 block, 64 tiles x 12 heads", 64, 12, 256, 256, 64, True),
         ("SD self-attn 64x64 latents (CFG batch 2)", 2, 8, 4096, 4096, 40, False), ("SD self-attn 32x32", 2, 8, 1024, 1024, 80, False),
         ("SD self-attn 16x16", 2, 8, 256, 256, 160, False), ("SD cross-attn 4096 x 77", 2, 8, 4096, 77, 40, False),
         # the same 4096 x 4096 problem at the head dims where the matrix work per score catches up with the softmax's vector work (round 3)
         ("self-attn 4096 x 4096, d 64", 2, 8, 4096, 4096, 64, False), ("self-attn 4096 x 4096, d 128", 2, 8, 4096, 4096, 128, False),
         ("self-attn 4096 x 4096, d 160", 2, 8, 4096, 4096, 160, False)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else None
reps = int(os.environ.get("REPS", "20"))
eng = get_engine("cuda")
for idx, (name, B, H, Lq, Lk, D, norm) in enumerate(CASES):
    if only is not None and idx != only:
        continue
    g = torch.Generator(device="cuda").manual_seed(idx)
    q, k, v = (torch.randn(B, H, L, D, device="cuda", generator=g) for L in (Lq, Lk, Lk))
    attention(q, k, v, normalize=norm)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        attention(q, k, v, normalize=norm)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / reps
    useful = 4.0 * B * H * Lq * Lk * D
    Dp, Dm = (D + 15) // 16 * 16, (D + 31) // 32 * 32   # (round 6: where D % 16 != 0 one padding channel / V^T row carries the softmax's reference point / row sum)
    Lkp, Lqp = (Lk + 63) // 64 * 64, (Lq + 127) // 128 * 128
    issued = 2.0 * B * H * Lqp * Lkp * (Dp + Dm)
    print(f"CASE {idx} | {name} | B{B} H{H} {Lq}x{Lk} d{D} | useful GFLOP {useful / 1e9:.3f} | issued MFMA GFLOP {issued / 1e9:.3f} | wall per call (pack + kernel + sync) {dt * 1e6:.1f} us")
