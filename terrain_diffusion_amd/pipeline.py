"""InfiniteDiffusion latent stage on the engine: the lazy, unbounded, multi-phase counterpart of the bounded samplers.

Mirrors the structure of WorldPipeline._build_latent_stage / _latent_inference (terrain_diffusion/inference/world_pipeline.py:
1052-1131, 1133-1203): a chain of InfiniteTensors, one per trig-flow phase, windows of `tile` stride `tile//2`, every window output
packed as (C+1, tile, tile) = (sample * w, w); a phase reads the blended (summed) previous phase through `args_windows`, divides by
the weight channel and re-noises with the portable field seeded `seed + 5819 + phase` at absolute coordinates.  All missing
windows of a request are batched through the U-Net (`batch_size`), noise / U-Net / consistency update run in the HIP engine.
The conditioning source is a callback (the reference's coarse stage and synthetic-map generator are out of scope).
"""
import math

import torch

from ._lib import lib, check
from .engine import ptr
from .infinite_tensor import InfiniteTensor, MemoryTileStore, TensorWindow
from . import noise as _noise
from .sampling import _linear_weight_window


def build_latent_stage(model, sigma_data=0.5, sigma_max=80.0, *, seed, cond_fn, intermediate_ts=(math.atan(0.35 / 0.5),), tile=64, channels=5,
                       batch_size=16, tile_store=None, tensor_prefix="latents"):
    """Returns the final-phase InfiniteTensor of shape (channels+1, None, None).
    cond_fn(ctxs) -> (len(ctxs), cond_dim) conditioning vectors for window indices ctxs = [(0, i, j), ...]."""
    stride = tile // 2
    store = tile_store if tile_store is not None else MemoryTileStore()
    dev = model.device
    w = _linear_weight_window(tile, dev)[0, 0].cpu()
    win = TensorWindow(size=(channels + 1, tile, tile), stride=(channels + 1, stride, stride))
    t_init = float(torch.atan(torch.tensor(sigma_max, dtype=torch.float32) / sigma_data))
    ts = (t_init,) + tuple(float(torch.tensor(t, dtype=torch.float32)) for t in intermediate_ts)

    def make_f(phase, t):
        def f(ctxs, prevs=None):
            n = len(ctxs)
            origins = [(c[1] * stride, c[2] * stride) for c in ctxs]
            z = _noise.gaussian_noise_patches(seed + 5819 + phase, origins, tile, tile, channels=channels, tile_h=tile, tile_w=tile, device=dev)
            sample = None
            if prevs is not None:  # (C+1, tile, tile) un-normalised sums -> sample * sigma_data (world_pipeline.py:1078)
                sample = torch.stack([(p[:-1] / p[-1:]) * sigma_data for p in prevs]).to(dev).contiguous()
            cond = torch.as_tensor(cond_fn(ctxs), dtype=torch.float32).to(dev).contiguous()
            out = torch.empty_like(z)
            check(lib().td_sample_consistency(model._h, n, tile, tile, float(t), float(sigma_data), ptr(sample), ptr(z), ptr(cond), ptr(out)))
            out = out.cpu() / sigma_data  # world_pipeline.py:1129
            return [torch.cat([o * w[None], w[None]], dim=0) for o in out]
        return f

    lat = InfiniteTensor((channels + 1, None, None), make_f(0, ts[0]), win, tile_store=store, tensor_id=f"{tensor_prefix}_phase0", batch_size=batch_size)
    for k, t in enumerate(ts[1:], 1):
        lat = InfiniteTensor((channels + 1, None, None), make_f(k, t), win, args=(lat,), args_windows=(win,), tile_store=store,
                             tensor_id=f"{tensor_prefix}_phase{k}", batch_size=batch_size)
    return lat
