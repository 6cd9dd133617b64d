"""Host helpers of the annotated InfiniteDiffusion panorama (annotated_infinite_panorama.py:57-102), restated:
1-D tiled deterministic noise (numpy SeedSequence -> PCG64DXSM, stays on the host by design: SURVEY.md a4), the 1-D linear blend
kernel and the phase partition of a descending timestep list.  The SD-v1.5 U-Net/VAE of that script live in `diffusers` and are
out of scope (parity unpinned); these helpers + infinite_tensor.py are the plumbing around a user-supplied denoiser."""
import numpy as np
import torch


def tiled_gaussian_noise(seed, x0, width, channels=4, height=64, tile=256):
    out = np.empty((channels, height, width), dtype=np.float32)
    first_tx, last_tx = x0 // tile, (x0 + width - 1) // tile
    for tx in range(first_tx, last_tx + 1):
        tile_x0 = tx * tile
        ox0, ox1 = max(x0, tile_x0), min(x0 + width, tile_x0 + tile)
        ss = np.random.SeedSequence(np.array([seed, tx & 0xFFFFFFFF], dtype=np.uint32))
        rng = np.random.Generator(np.random.PCG64DXSM(ss))
        tile_noise = rng.standard_normal((channels, height, tile), dtype=np.float32)
        out[:, :, ox0 - x0:ox1 - x0] = tile_noise[:, :, ox0 - tile_x0:ox1 - tile_x0]
    return out


def linear_kernel(height, width):
    x = torch.arange(width, dtype=torch.float32)
    mid = (width - 1) / 2
    w = 1 - 0.999 * torch.abs(x - mid) / mid
    return w[None, :].expand(height, -1).contiguous()


def build_timestep_ranges(all_timesteps, thresholds):
    thresholds = sorted(thresholds, reverse=True)
    if not thresholds:
        return [all_timesteps]
    ranges, prev = [], None
    for t in thresholds:
        r = all_timesteps[all_timesteps >= t] if prev is None else all_timesteps[(all_timesteps >= t) & (all_timesteps < prev)]
        if len(r) > 0:
            ranges.append(r)
        prev = t
    tail = all_timesteps[all_timesteps < thresholds[-1]]
    if len(tail) > 0:
        ranges.append(tail)
    return ranges


def normalize(weighted, clamp=1e-6):
    """annotated_infinite_panorama.py:145-146 (weight channel clamped at 1e-6)."""
    return weighted[:-1] / weighted[-1:].clamp(min=clamp)


def pack(values_chw, weight_hw):
    """annotated_infinite_panorama.py:148-150."""
    return torch.cat([values_chw * weight_hw[None], weight_hw[None]], dim=0)
