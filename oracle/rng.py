"""ORACLE: portable RNG + tile-seeded noise field (CPU).

Restates terrain_diffusion/inference/portable_rng.py:22-89 and
terrain_diffusion/inference/world_pipeline.py:58-115.  Two forms: pure-Python big-int (small
cases, readable) and the C restatement in oracle/csrc/portable_rng.c (fast, via ctypes).
"""
import ctypes
import math
import os

import numpy as np

from . import build as _build

M64 = 0xFFFFFFFFFFFFFFFF
PCG_MULT = 6364136223846793005
PCG_INC = 1442695040888963407


# ---------------------------------------------------------------- pure Python (small cases)
def pcg_next_py(state: int):
    """portable_rng.py:22-28 — one LCG advance, XSH-RR output of the NEW state."""
    state = (state * PCG_MULT + PCG_INC) & M64
    x = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
    rot = state >> 59
    return state, ((x >> rot) | (x << ((32 - rot) & 31))) & 0xFFFFFFFF


def next_seed_py(seed: int) -> int:
    """portable_rng.py:31-42 (explicit non-zero seed only)."""
    s = seed & M64
    s, lo = pcg_next_py(s)
    s, hi = pcg_next_py(s)
    return ((hi << 32) | lo) & M64


def standard_normal_py(seed: int, n: int) -> np.ndarray:
    """portable_rng.py:56-74 — Marsaglia polar in f64, stored to f32."""
    out = np.empty(n, dtype=np.float32)
    s = seed & M64
    i = 0
    inv = 1.0 / 4294967296.0
    while i < n:
        s, u1 = pcg_next_py(s)
        s, u2 = pcg_next_py(s)
        v1 = 2.0 * (float(u1) + 1.0) * inv - 1.0
        v2 = 2.0 * (float(u2) + 1.0) * inv - 1.0
        r = v1 * v1 + v2 * v2
        if 0.0 < r < 1.0:
            f = math.sqrt(-2.0 * math.log(r) / r)
            out[i] = v1 * f
            i += 1
            if i < n:
                out[i] = v2 * f
                i += 1
    return out


def tile_seed_py(base_seed: int, ty: int, tx: int) -> int:
    """world_pipeline.py:58-63."""
    h = (int(base_seed) & M64) * 0x9E3779B9
    h = (h + (int(ty) & 0xFFFFFFFF)) & M64
    h = (h * 0x9E3779B9 + (int(tx) & 0xFFFFFFFF)) & M64
    return h


# ---------------------------------------------------------------- C restatement via ctypes
_lib = None


def _c():
    global _lib
    if _lib is None:
        path = _build.build()
        lib = ctypes.CDLL(path)
        lib.orc_pcg_stream.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64]
        lib.orc_next_seed.argtypes = [ctypes.c_uint64]
        lib.orc_next_seed.restype = ctypes.c_uint64
        lib.orc_fill_standard_normal_f32.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64]
        lib.orc_fill_standard_normal_f64.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64]
        lib.orc_tile_seed.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64]
        lib.orc_tile_seed.restype = ctypes.c_uint64
        lib.orc_gaussian_noise_patch.argtypes = [ctypes.c_uint64] + [ctypes.c_int64] * 7 + [ctypes.c_void_p]
        lib.orc_gaussian_noise_patch.restype = ctypes.c_int
        _lib = lib
    return _lib


def pcg_stream(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    _c().orc_pcg_stream(seed & M64, out.ctypes.data, n)
    return out


def next_seed(seed: int) -> int:
    return int(_c().orc_next_seed(seed & M64))


def standard_normal(seed: int, size, dtype=np.float32) -> np.ndarray:
    out = np.empty(size, dtype=dtype)
    if out.size == 0:
        return out
    if out.dtype == np.float32:
        _c().orc_fill_standard_normal_f32(int(seed) & M64, out.ctypes.data, out.size)
    elif out.dtype == np.float64:
        _c().orc_fill_standard_normal_f64(int(seed) & M64, out.ctypes.data, out.size)
    else:
        raise TypeError(dtype)
    return out


def tile_seed(base_seed: int, ty: int, tx: int) -> int:
    return int(_c().orc_tile_seed(int(base_seed) & M64, int(ty), int(tx)))


def gaussian_noise_patch(base_seed, y0, x0, h, w, channels=1, tile_h=256, tile_w=256) -> np.ndarray:
    """world_pipeline.py:66-115 — (C,h,w) window of the infinite tile-seeded noise field."""
    out = np.empty((channels, h, w), dtype=np.float32)
    rc = _c().orc_gaussian_noise_patch(int(base_seed) & M64, y0, x0, h, w, channels, tile_h, tile_w, out.ctypes.data)
    if rc != 0:
        raise MemoryError
    return out
