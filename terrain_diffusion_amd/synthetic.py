"""Synthetic weights / conditioning for benchmarks and smoke runs (no checkpoints exist offline).

Weights are portable-RNG standard normals seeded by FNV-1a(name) ^ seed — generated on the GPU by the engine's own
noise kernel — with non-zero gains (fresh EDMUnet2D gains are 0, which would zero the output: SURVEY.md Q1).
The same recipe is restated on the CPU in oracle/unet.py:synth_state_dict for the parity tests.
"""
import math

import torch

from . import noise as _noise

M64 = 0xFFFFFFFFFFFFFFFF


def _name_seed(name: str, seed: int) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & M64
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & M64 or 1


def synthetic_state_dict(model, seed=1234, out_gain=1.0, emb_gain=0.5):
    sd = {}
    for name, shape in model.expected_parameters().items():
        if name == "out_gain":
            sd[name] = torch.tensor(float(out_gain))
        elif name.endswith("emb_gain"):
            sd[name] = torch.tensor(float(emb_gain))
        elif name == "noise_fourier.freqs":
            half = shape[0]
            sd[name] = torch.exp(torch.arange(half) * -(math.log(10) / (half - 1)))  # mp_layers.py:89-94
        else:
            sd[name] = _noise.standard_normal(_name_seed(name, seed), shape, device=model.device, as_torch=True).cpu()
    return sd


def synthetic_cond_grid(n_ty, n_tx, seed=0xC0DE, device="cuda"):
    """(1, 7, n_ty+3, n_tx+3) standard normals."""
    return _noise.standard_normal(seed, (1, 7, n_ty + 3, n_tx + 3), device=device, as_torch=True).cpu()
