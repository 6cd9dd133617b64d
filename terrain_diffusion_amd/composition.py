"""Output composition after the decoder: decoder residual + latent low-frequency channel -> elevation in metres, coarse map -> climate.

Mirrors WorldPipeline._compute_elev / _compute_climate / get (terrain_diffusion/inference/world_pipeline.py:1276-1384) and the Laplacian
pyramid helpers they call (terrain_diffusion/data/laplacian_encoder.py:6-137).  The image operators run in the engine
(csrc/compose_kernels.hip through td_resample2d / td_residual_plus / td_elev_finish); this module builds their tap tables in the
arithmetic of the operators the reference reaches through torchvision:
  * TF.resize(..., BILINEAR) on tensors == torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=True);
    anti-aliasing only acts when shrinking (triangle filter stretched by the scale factor),
  * TF.gaussian_blur == reflect padding + correlation with the normalised sampled Gaussian of torchvision's `_get_gaussian_kernel1d`.
torchvision itself is not installed here, so those two operators are PARITY-UNPINNED against the reference; they are pinned against torch's own
F.interpolate / conv2d in tests (oracle/compose.py), which is what torchvision calls.
"""
import ctypes as C

import functools

import numpy as np
import torch

from ._lib import lib, check
from .engine import ptr

LOWFREQ_MEAN, LOWFREQ_STD = -31.4, 38.6            # world_pipeline.py:1279-1280
ELEV_SIGMA = 5                                     # world_pipeline.py:1284


# ------------------------------------------------------------------------------------------------ tap tables (host, tiny)
@functools.lru_cache(maxsize=256)   # pure functions of the sizes: a request stream asks for the same few tables over and over (23 ms of Python per cascade step)
def bilinear_taps(n_in, n_out):
    """torch upsample_bilinear2d, align_corners=False: two taps per output index, fp32 source coordinates clamped at 0."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = np.maximum(scale * (dst + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1) - l1).astype(np.float32)
    return np.stack([i0, i1], 1).astype(np.int32), np.stack([l0, l1], 1).astype(np.float32)


@functools.lru_cache(maxsize=256)   # pure functions of the sizes: a request stream asks for the same few tables over and over (23 ms of Python per cascade step)
def bilinear_aa_taps(n_in, n_out):
    """torch _upsample_bilinear2d_aa (what TF.resize(..., antialias=True) runs): triangle filter of half-width max(scale, 1)."""
    scale = n_in / n_out
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    rows = []
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), n_in) - xmin
        w = np.array([max(0.0, 1.0 - abs((j + xmin - center + 0.5) * invscale)) for j in range(xsize)], dtype=np.float64)
        w = (w / w.sum()).astype(np.float32)
        rows.append((xmin, w))
    K = max(len(w) for _, w in rows)
    idx = np.zeros((n_out, K), np.int32)
    wts = np.zeros((n_out, K), np.float32)
    for i, (xmin, w) in enumerate(rows):
        idx[i, :len(w)] = xmin + np.arange(len(w))
        idx[i, len(w):] = xmin
        wts[i, :len(w)] = w
    return idx, wts


@functools.lru_cache(maxsize=256)   # pure functions of the sizes: a request stream asks for the same few tables over and over (23 ms of Python per cascade step)
def gaussian_taps(n, sigma):
    """torchvision gaussian_blur: kernel_size = int(sigma*2)//2*2 + 1 samples of exp(-x^2 / 2 sigma^2) on linspace(-h, h), normalised, reflect padding."""
    k = int(sigma * 2) // 2 * 2 + 1
    half = (k - 1) * 0.5
    x = torch.linspace(-half, half, steps=k)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    ker = (pdf / pdf.sum()).numpy().astype(np.float32)
    pad = k // 2
    pos = np.arange(n)[:, None] + np.arange(k)[None, :] - pad
    pos = np.where(pos < 0, -pos, pos)
    pos = np.where(pos >= n, 2 * (n - 1) - pos, pos)
    return pos.astype(np.int32), np.broadcast_to(ker[None, :], (n, k)).copy()


def resample(engine, x, taps_y, taps_x):
    """x: (C,H,W) or (H,W) fp32 device tensor -> resampled by the row / column tap tables, in the engine."""
    squeeze = x.ndim == 2
    x3 = (x[None] if squeeze else x).contiguous().float()
    (iy, wy), (ix, wx) = taps_y, taps_x
    iy, wy, ix, wx = (np.ascontiguousarray(a) for a in (iy, wy, ix, wx))
    out = torch.empty((x3.shape[0], iy.shape[0], ix.shape[0]), dtype=torch.float32, device=x3.device)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    check(lib().td_resample2d(engine._h, ptr(x3), x3.shape[0], x3.shape[1], x3.shape[2], iy.shape[0], ix.shape[0], vp(iy), vp(wy), iy.shape[1],
                              vp(ix), vp(wx), ix.shape[1], ptr(out)))
    return out[0] if squeeze else out


def resize_bilinear(engine, x, size, antialias=True):
    """TF.resize(x, size, BILINEAR) for a device tensor (..., H, W)."""
    H, W = x.shape[-2:]
    ty = bilinear_aa_taps(H, size[0]) if (antialias and size[0] < H) else bilinear_taps(H, size[0])
    tx = bilinear_aa_taps(W, size[1]) if (antialias and size[1] < W) else bilinear_taps(W, size[1])
    return resample(engine, x, ty, tx)


def gaussian_blur(engine, x, sigma):
    return resample(engine, x, gaussian_taps(x.shape[-2], sigma), gaussian_taps(x.shape[-1], sigma))


def pad_linear_extrapolation(x):
    """laplacian_encoder.py:6-39 on a (H, W) device tensor: one ring of linearly extrapolated values."""
    h, w = x.shape[-2:]
    top, bot = (2 * x[0:1] - x[1:2], 2 * x[-1:] - x[-2:-1]) if h > 1 else (x[0:1], x[-1:])
    x = torch.cat([top, x, bot], dim=-2)
    left, right = (2 * x[:, 0:1] - x[:, 1:2], 2 * x[:, -1:] - x[:, -2:-1]) if w > 1 else (x[:, 0:1], x[:, -1:])
    return torch.cat([left, x, right], dim=-1)


def resize_extrapolated(engine, x, size):
    """laplacian_encoder.py:41-60: resize the linearly padded image and crop the padding's share."""
    th, tw = size
    h, w = x.shape[-2:]
    sh, sw = th / h, tw / w
    out = resize_bilinear(engine, pad_linear_extrapolation(x), (int(round(th + 2 * sh)), int(round(tw + 2 * sw))))
    ph, pw = int(round(sh)), int(round(sw))
    return out[ph:ph + th, pw:pw + tw]


# ------------------------------------------------------------------------------------------------ elevation / climate
@torch.no_grad()
def compute_elev(engine, residual, latents, i1, j1, i2, j2, scale, residual_mean, residual_std):
    """WorldPipeline._compute_elev (world_pipeline.py:1276-1313): (h, w) elevation in metres for the pixel box [i1,i2) x [j1,j2).
    residual: decoder stage tensor (2, None, None) (packed residual * w, w); latents: latent stage tensor (6, None, None)."""
    sigma = ELEV_SIGMA
    ksize = (int(sigma * 2) // 2) * 2 + 1
    pad_hr = (ksize // 2 + 1) * scale
    pi1, pj1 = ((i1 - pad_hr) // scale) * scale, ((j1 - pad_hr) // scale) * scale
    pi2, pj2 = -((-(i2 + pad_hr)) // scale) * scale, -((-(j2 + pad_hr)) // scale) * scale
    dev = torch.device("cuda", engine.device_id)
    packed = torch.as_tensor(residual[:, pi1:pi2, pj1:pj2]).to(dev, torch.float32).contiguous()
    lat = torch.as_tensor(latents[:, pi1 // scale:pi2 // scale, pj1 // scale:pj2 // scale]).to(dev, torch.float32)
    lowfreq = (lat[4] / lat[-1]) * LOWFREQ_STD + LOWFREQ_MEAN
    Hp, Wp = packed.shape[-2:]
    # laplacian_denoise: decode with extrapolated upsampling, re-encode the low band (anti-aliased shrink + blur)  (laplacian_encoder.py:134-137)
    up0 = resize_extrapolated(engine, lowfreq, (Hp, Wp)).contiguous()
    decoded = torch.empty((Hp, Wp), dtype=torch.float32, device=dev)
    check(lib().td_residual_plus(engine._h, ptr(packed), ptr(up0), Hp, Wp, float(residual_mean), float(residual_std), ptr(decoded)))
    n_low = lowfreq.shape[-1]
    low = resize_bilinear(engine, decoded, _resize_int_size(Hp, Wp, n_low))
    low = gaussian_blur(engine, low, sigma)
    # laplacian_decode with the denoised low band, crop, signed square  (world_pipeline.py:1306-1312)
    up1 = resize_bilinear(engine, low, (Hp, Wp)).contiguous()
    h, w = i2 - i1, j2 - j1
    out = torch.empty((h, w), dtype=torch.float32, device=dev)
    check(lib().td_elev_finish(engine._h, ptr(packed), ptr(up1), Hp, Wp, i1 - pi1, j1 - pj1, h, w, float(residual_mean), float(residual_std), ptr(out)))
    return out


def _resize_int_size(h, w, size):
    """TF.resize with an int size: the SMALLER edge becomes `size`, the other keeps the aspect ratio (torchvision _compute_resized_output_size)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def local_baseline_temperature(T, e, win=3, beta_clip=(-0.012, 0.0), fallback_beta=-0.0065, eps=1e-6, fallback_threshold=0.3):
    """postprocessing.py:262-324: windowed regression of temperature on elevation over land -> (sea-level baseline, lapse rate), (H-win+1, W-win+1)."""
    F = torch.nn.functional
    T, e = T[None, None], e[None, None]
    wgt = (e > 0).float()

    def wavg(x):
        return F.avg_pool2d(x * wgt, win, stride=1, padding=0) / (F.avg_pool2d(wgt, win, stride=1, padding=0) + eps)
    sum_w = F.avg_pool2d(wgt, win, stride=1, padding=0)
    mu_T, mu_e, mu_e2, mu_eT = wavg(T), wavg(e), wavg(e * e), wavg(e * T)
    var_e = mu_e2 - mu_e ** 2
    beta = (mu_eT - mu_e * mu_T) / (var_e + eps)
    beta = torch.where((var_e < 1.0) | (sum_w < fallback_threshold), float(fallback_beta), beta)   # scalar overload: no host-to-device copy (and no sync)
    beta = torch.clamp(beta, beta_clip[0], beta_clip[1])
    pad = (win - 1) // 2
    T_sea = T[:, :, pad:-pad, pad:-pad] - beta * e[:, :, pad:-pad, pad:-pad]
    return T_sea[0, 0], beta[0, 0]


@torch.no_grad()
def compute_climate(coarse, i1, j1, i2, j2, elev, scale, engine=None):
    """WorldPipeline._compute_climate (world_pipeline.py:1315-1365): (5, h, w) = realistic temperature, three climate channels, lapse rate.
    The windowed regression runs on a few hundred coarse cells (device torch pooling ops); everything per output pixel is one HIP kernel
    (td_climate_finish)."""
    S = 32 * scale
    ci1, cj1 = i1 // S, j1 // S
    ci2, cj2 = -((-i2) // S), -((-j2) // S)
    win = 15
    cpad = (win - 1) // 2 + 1
    dev = elev.device
    cinit = torch.as_tensor(coarse[:, ci1 - cpad:ci2 + cpad, cj1 - cpad:cj2 + cpad]).to(dev, torch.float32)
    cmap = cinit[:-1] / cinit[-1:]
    celev = torch.sign(cmap[0]) * torch.square(torch.clamp(cmap[0], min=0))
    base, beta = local_baseline_temperature(cmap[2], celev, win=win, fallback_threshold=0.02)
    central = cmap[:, win // 2:-(win // 2), win // 2:-(win // 2)]
    Hs, Ws = base.shape[-2:]
    # per-pixel half (grid, grid_sample of the features, lapse-rate temperature, stack): one pass of climate_finish_kernel over the request
    feats = torch.stack([base, beta, central[3], central[4], central[5]]).contiguous()
    elev = elev.to(torch.float32).contiguous()
    h, w = i2 - i1, j2 - j1
    out = torch.empty((5, h, w), dtype=torch.float32, device=dev)
    from .engine import get_engine
    eng = engine if engine is not None else get_engine(dev)
    check(lib().td_climate_finish(eng._h, ptr(feats), int(Hs), int(Ws), ptr(elev), int(i1), int(j1), int(h), int(w), float(S), int(ci1), int(cj1), ptr(out)))
    return out
