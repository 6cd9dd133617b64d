"""A/B of the LDS-DMA 1x1-segment path of conv_glds (engine option glds_dma1x1): the network output must not change by a bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from oracle.unet import BASE_CONFIG, synth_state_dict
from oracle import rng

eng = td.engine.get_engine("cuda")
cfg = dict(BASE_CONFIG)
m = td.EDMUnet2D(**cfg, dtype="bf16").load_state_dict(synth_state_dict(cfg, seed=1234))
for n in (64, 16, 3, 1):
    x = torch.from_numpy(rng.standard_normal(7, (n, 5, 64, 64))).cuda()
    c = torch.from_numpy(rng.standard_normal(8, (n, 58))).cuda()
    t = torch.full((n,), 1.1)
    outs = {}
    for o in (0, 1):
        eng.set_option("glds_dma1x1", o)
        outs[o] = m(x, t, [c]).clone()
    same = torch.equal(outs[0], outs[1])
    print(f"batch {n}: dma1x1 output bit-identical to the register path: {same}; finite: {bool(torch.isfinite(outs[1]).all())}; max|diff| {float((outs[0] - outs[1]).abs().max()):.3e}")
