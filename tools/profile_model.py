"""Kernel-time breakdown for any of the three pipeline models: python tools/profile_model.py base|coarse|decoder N H [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from terrain_diffusion_amd.synthetic import synthetic_state_dict
from terrain_diffusion_amd.engine import get_engine

CFG = {
    "base": dict(image_size=512, in_channels=5, out_channels=5, model_channels=192, model_channel_mults=[1, 2, 3, 4], layers_per_block=3, attn_resolutions=[8, 16],
                 midblock_attention=True, concat_balance=0.5, conditional_inputs=[["tensor", 58, 1.0]], fourier_scale="pos"),
    "coarse": dict(image_size=16, in_channels=11, out_channels=6, model_channels=128, model_channel_mults=[1], layers_per_block=2, attn_resolutions=[],
                   midblock_attention=False, concat_balance=0.5, conditional_inputs=[["float", 64, 0.2]] * 5, fourier_scale="pos"),
    "decoder": dict(image_size=512, in_channels=5, out_channels=1, model_channels=64, model_channel_mults=[1, 2, 3, 4], layers_per_block=3, attn_resolutions=[],
                    midblock_attention=False, concat_balance=0.5, conditional_inputs=[], fourier_scale="pos"),
}
GFLOP = {"base": 193.654, "coarse": 21.511, "decoder": 1375.749}   # per forward at 64^2 / 64^2 / 512^2
which, n, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
for kv in os.environ.get('TD_OPTS', '').split(','):
    if kv:
        k, v = kv.split('='); get_engine("cuda").set_option(k, int(v))
cfg = CFG[which]
m = td.EDMUnet2D(**cfg, dtype=dtype)
m.load_state_dict(synthetic_state_dict(m, seed=1))
eng = m.engine
x = torch.randn(n, cfg["in_channels"], H, H, device="cuda")
conds = [torch.randn(n, c[1]) if c[0] == "tensor" else torch.randn(n) for c in cfg["conditional_inputs"]]
t = torch.full((n,), 1.1)
m(x, t, conds)
eng.set_option("profile", 1); eng.profile_read(reset=True)
reps = 3
for _ in range(reps):
    m(x, t, conds)
rows = eng.profile_ops(); conv_ms, conv_n, other_ms, other_n = eng.profile_read()
tot = sum(r[1] for r in rows) / reps
base_hw = 512 if which == "decoder" else 64
gf = GFLOP[which] * n * (H / base_hw) ** 2
print(f"{which} batch {n} {H}x{H} {dtype}: {tot:.3f} ms kernel time per forward, {gf / tot:.1f} TFLOP/s ({conv_n // reps} conv launches)")
for r in sorted(rows, key=lambda r: -r[1])[:int(os.environ.get("TD_TOP", "14"))]:
    print(f"{r[1] / reps * 1e3:9.1f} us  {r[0]}")
