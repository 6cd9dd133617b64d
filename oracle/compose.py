"""ORACLE (test infrastructure): output composition after the decoder, restated on CPU torch.

Follows terrain_diffusion/data/laplacian_encoder.py:6-137 (pad_linear_extrapolation, resize_extrapolated, laplacian_encode / decode / denoise),
terrain_diffusion/inference/world_pipeline.py:1276-1365 (_compute_elev, _compute_climate) and
terrain_diffusion/inference/postprocessing.py:262-324 (local_baseline_temperature_torch; pinned to the reference's own output through
tests/golden/compose.npz).  PARITY UNPINNED for the two torchvision operators (torchvision is not installed in the build container):
TF.resize(BILINEAR) is restated as F.interpolate(mode="bilinear", align_corners=False, antialias=True) -- the call torchvision>=0.19 makes for
tensors -- and TF.gaussian_blur as reflect padding + conv2d with torchvision's sampled, normalised Gaussian.
"""
import torch
import torch.nn.functional as F

LOWFREQ_MEAN, LOWFREQ_STD = -31.4, 38.6


def tf_resize(x, size):
    """TF.resize(x, size, BILINEAR) for (..., H, W); int size = smaller edge."""
    lead = x.shape[:-2]
    h, w = x.shape[-2:]
    if isinstance(size, int):
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long_ / short)
        size = (new_long, new_short) if w <= h else (new_short, new_long)
    y = F.interpolate(x.reshape(1, -1, h, w).float(), size=tuple(size), mode="bilinear", align_corners=False, antialias=True)
    return y.reshape(*lead, *size)


def tf_gaussian_blur(x, sigma):
    k = int(sigma * 2) // 2 * 2 + 1
    half = (k - 1) * 0.5
    g = torch.linspace(-half, half, steps=k)
    pdf = torch.exp(-0.5 * (g / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    lead = x.shape[:-2]
    h, w = x.shape[-2:]
    y = F.pad(x.reshape(1, -1, h, w).float(), [k // 2] * 4, mode="reflect")
    c = y.shape[1]
    y = F.conv2d(y, k2[None, None].expand(c, 1, k, k), groups=c)
    return y.reshape(*lead, h, w)


def pad_linear_extrapolation(x):
    h, w = x.shape[-2:]
    top, bot = (2 * x[..., 0:1, :] - x[..., 1:2, :], 2 * x[..., -1:, :] - x[..., -2:-1, :]) if h > 1 else (x[..., 0:1, :], x[..., -1:, :])
    x = torch.cat([top, x, bot], dim=-2)
    left, right = (2 * x[..., :, 0:1] - x[..., :, 1:2], 2 * x[..., :, -1:] - x[..., :, -2:-1]) if w > 1 else (x[..., :, 0:1], x[..., :, -1:])
    return torch.cat([left, x, right], dim=-1)


def resize_extrapolated(x, size):
    th, tw = size
    h, w = x.shape[-2:]
    sh, sw = th / h, tw / w
    out = tf_resize(pad_linear_extrapolation(x), (int(round(th + 2 * sh)), int(round(tw + 2 * sw))))
    ph, pw = int(round(sh)), int(round(sw))
    return out[..., ph:ph + th, pw:pw + tw]


def laplacian_encode(x, downsample_size, sigma):
    low = tf_gaussian_blur(tf_resize(x, downsample_size), sigma)
    return x - tf_resize(low, x.shape[-2:]), low


def laplacian_decode(residual, lowres, extrapolate=False):
    up = resize_extrapolated(lowres, residual.shape[-2:]) if extrapolate else tf_resize(lowres, residual.shape[-2:])
    return residual + up


def laplacian_denoise(residual, lowres, sigma):
    decoded = laplacian_decode(residual, lowres, extrapolate=True)
    _, new_low = laplacian_encode(decoded, lowres.shape[-1], sigma)
    return residual, new_low


def compute_elev(residual_map, latents, i1, j1, i2, j2, scale, residual_mean, residual_std):
    """world_pipeline.py:1276-1313 with `residual_map` / `latents` anything sliceable that returns packed (C+1, h, w) sums."""
    sigma = 5
    ksize = (int(sigma * 2) // 2) * 2 + 1
    pad_hr = (ksize // 2 + 1) * scale
    pi1, pj1 = ((i1 - pad_hr) // scale) * scale, ((j1 - pad_hr) // scale) * scale
    pi2, pj2 = -((-(i2 + pad_hr)) // scale) * scale, -((-(j2 + pad_hr)) // scale) * scale
    r = torch.as_tensor(residual_map[:, pi1:pi2, pj1:pj2]).float()
    residual_p = (r[0] / r[1]) * residual_std + residual_mean
    lat = torch.as_tensor(latents[:, pi1 // scale:pi2 // scale, pj1 // scale:pj2 // scale]).float()
    lowfreq_p = (lat[:-1] / lat[-1:])[4] * LOWFREQ_STD + LOWFREQ_MEAN
    residual_p, lowfreq_p = laplacian_denoise(residual_p, lowfreq_p, sigma)
    elev_p = laplacian_decode(residual_p, lowfreq_p)
    oi, oj = i1 - pi1, j1 - pj1
    e = elev_p[oi:oi + i2 - i1, oj:oj + j2 - j1]
    return torch.sign(e) * torch.square(e)


def local_baseline_temperature(T, e, win=3, beta_clip=(-0.012, 0.0), fallback_beta=-0.0065, eps=1e-6, fallback_threshold=0.3):
    T, e = T[None, None], e[None, None]
    w = (e > 0).float()

    def wavg(x):
        return F.avg_pool2d(x * w, win, stride=1, padding=0) / (F.avg_pool2d(w, win, stride=1, padding=0) + eps), F.avg_pool2d(w, win, stride=1, padding=0)
    mu_T, sum_w = wavg(T)
    mu_e, _ = wavg(e)
    mu_e2, _ = wavg(e * e)
    mu_eT, _ = wavg(e * T)
    var_e = mu_e2 - mu_e ** 2
    beta = (mu_eT - mu_e * mu_T) / (var_e + eps)
    beta = torch.where((var_e < 1.0) | (sum_w < fallback_threshold), torch.tensor(fallback_beta), beta)
    beta = torch.clamp(beta, beta_clip[0], beta_clip[1])
    pad = (win - 1) // 2
    return (T[:, :, pad:-pad, pad:-pad] - beta * e[:, :, pad:-pad, pad:-pad])[0, 0], beta[0, 0]


def compute_climate(coarse, i1, j1, i2, j2, elev, scale):
    S = 32 * scale
    ci1, cj1, ci2, cj2 = i1 // S, j1 // S, -((-i2) // S), -((-j2) // S)
    win = 15
    cpad = (win - 1) // 2 + 1
    cinit = torch.as_tensor(coarse[:, ci1 - cpad:ci2 + cpad, cj1 - cpad:cj2 + cpad]).float()
    cmap = cinit[:-1] / cinit[-1:]
    celev = torch.sign(cmap[0]) * torch.square(torch.maximum(torch.zeros_like(cmap[0]), cmap[0]))
    base, beta = local_baseline_temperature(cmap[2], celev, win=win, fallback_threshold=0.02)
    central = cmap[:, win // 2:-(win // 2), win // 2:-(win // 2)]
    Hs, Ws = base.shape[-2:]
    ii, jj = torch.meshgrid(torch.arange(i1, i2), torch.arange(j1, j2), indexing="ij")
    u = (ii + 0.5) / S - ci1 + 0.5
    v = (jj + 0.5) / S - cj1 + 0.5
    grid = torch.stack([(v + 0.5) * 2 / Ws - 1, (u + 0.5) * 2 / Hs - 1], dim=-1).unsqueeze(0)
    feats = torch.cat([base[None], beta[None], central], dim=0).unsqueeze(0)
    up = F.grid_sample(feats, grid, mode="bilinear", padding_mode="border", align_corners=False).squeeze(0)
    temp = up[0] + up[1] * torch.maximum(elev, torch.zeros_like(elev))
    return torch.stack([temp, up[5], up[6], up[7], up[1]])
