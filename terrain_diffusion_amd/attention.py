"""softmax(scale * Q K^T) V on the engine's MFMA flash kernel (csrc/attn_mfma.hip) -- the attention of UNetBlock.attn
(terrain_diffusion/models/unet_block.py:102-108) and, with normalize=False, the scaled-dot-product attention of the SD-v1.5 U-Net of the
panorama demo (annotated_infinite_panorama.py:109-134: torch.nn.functional.scaled_dot_product_attention inside diffusers)."""
import math

import torch

from ._lib import lib, check
from .engine import get_engine, ptr, f32


def attention(q, k, v, scale=None, normalize=False, device=None):
    """q: (B, H, Lq, D), k / v: (B, H, Lk, D) -> (B, H, Lq, D) fp32 on the engine's device.  scale defaults to 1 / sqrt(D)."""
    q, k, v = f32(q), f32(k), f32(v)
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    assert k.shape == (B, H, Lk, D) and v.shape == (B, H, Lk, D)
    eng = get_engine(device if device is not None else (q.device if q.is_cuda else "cuda"))
    dev = torch.device("cuda", eng.device_id)
    q, k, v = (t.to(dev).contiguous() for t in (q, k, v))
    out = torch.empty((B, H, Lq, D), dtype=torch.float32, device=dev)
    check(lib().td_attention(eng._h, ptr(q), ptr(k), ptr(v), B, H, Lq, Lk, D, float(scale if scale is not None else 1.0 / math.sqrt(D)), int(bool(normalize)), ptr(out)))
    return out
