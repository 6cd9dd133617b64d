"""Portable tile-seeded Gaussian noise on the GPU, reference surface:
terrain_diffusion/inference/portable_rng.py:83-89 (standard_normal) and
terrain_diffusion/inference/world_pipeline.py:58-115 (_tile_seed, gaussian_noise_patch).
The PCG64/32 integer stream and the accept/reject order are reproduced exactly by `noise_tiles_kernel`.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check
from .engine import get_engine, ptr

M64 = 0xFFFFFFFFFFFFFFFF


def _tile_seed(base_seed: int, ty: int, tx: int) -> int:
    return int(lib().td_tile_seed(int(base_seed) & M64, int(ty), int(tx)))


def next_seed(seed=None):
    """portable_rng.py:31-42: a new 64-bit seed from a parent seed, or from the clock when seed is None / 0.  Host integers only: two steps
    of the module's LCG (s <- s * 6364136223846793005 + 1442695040888963407 mod 2^64), each giving one 32-bit XSH-RR word of the NEW state."""
    state = (int(seed) & M64) if seed is not None else 0
    if state == 0:
        import time
        state = int(time.perf_counter_ns()) & M64
    words = []
    for _ in range(2):
        state = (state * 6364136223846793005 + 1442695040888963407) & M64
        x = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
        rot = state >> 59
        words.append(((x >> rot) | (x << ((32 - rot) & 31))) & 0xFFFFFFFF)
    return int(((words[1] << 32) | words[0]) & M64)


def standard_normal(seed, size, dtype=np.float32, device="cuda", as_torch=False):
    shape = (size,) if isinstance(size, int) else tuple(size)
    n = int(np.prod(shape)) if shape else 1
    eng = get_engine(device)
    out = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", eng.device_id))
    if n:
        check(lib().td_standard_normal(eng._h, int(seed) & M64, n, ptr(out)))
    out = out.reshape(shape)
    return out if as_torch else out.cpu().numpy().astype(dtype, copy=False)


def gaussian_noise_patches(base_seed, origins, h, w, channels=1, tile_h=256, tile_w=256, scale=1.0, device="cuda"):
    """Batched gaussian_noise_patch: origins = [(y0, x0), ...] -> device tensor (n, channels, h, w)."""
    eng = get_engine(device)
    org = np.ascontiguousarray(np.asarray(origins, dtype=np.int64).reshape(-1, 2))
    out = torch.empty((len(org), channels, h, w), dtype=torch.float32, device=torch.device("cuda", eng.device_id))
    check(lib().td_noise_patches(eng._h, int(base_seed) & M64, len(org), C.c_void_p(org.ctypes.data), h, w, channels, tile_h, tile_w, float(scale), ptr(out)))
    return out


def gaussian_noise_patch(base_seed, y0, x0, h, w, channels=1, tile_h=256, tile_w=256, dtype=np.float32, device="cuda"):
    """Reference signature (world_pipeline.py:66-76); returns a (C, h, w) numpy array like the reference."""
    return gaussian_noise_patches(base_seed, [(y0, x0)], h, w, channels, tile_h, tile_w, 1.0, device)[0].cpu().numpy().astype(dtype, copy=False)
