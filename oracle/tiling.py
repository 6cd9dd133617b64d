"""ORACLE: tile geometry, blend windows, conditioning vector and the bounded tiled samplers (CPU).

Restates terrain_diffusion/training/evaluation/__init__.py:3-22 (_linear_weight_window, _tile_starts),
terrain_diffusion/training/evaluation/sample_diffusion_base.py:11-48 (_process_cond_img), :115-168
(sample_base_diffusion tiled branch), :171-268 (sample_base_consistency), and the helpers of
annotated_infinite_panorama.py:76-102 (linear_kernel, build_timestep_ranges).
Noise is addressed by absolute canvas coordinates through the portable tile-seeded field
(world_pipeline.py:66-115) instead of torch.randn.
"""
import math

import numpy as np
import torch

from . import rng, schedule
from .unet import mp_concat_scales  # noqa: F401  (re-export for tests)


def linear_weight_window(size, dtype=torch.float32):
    s = size
    mid = (s - 1) / 2
    y, x = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    eps = 1e-3
    wy = 1 - (1 - eps) * torch.clamp(torch.abs(y - mid).to(dtype) / mid, 0, 1)
    wx = 1 - (1 - eps) * torch.clamp(torch.abs(x - mid).to(dtype) / mid, 0, 1)
    return wy * wx


def pano_linear_kernel(height, width):
    x = torch.arange(width, dtype=torch.float32)
    mid = (width - 1) / 2
    w = 1 - 0.999 * torch.abs(x - mid) / mid
    return w[None, :].expand(height, -1).contiguous()


def tile_starts(length, tile_size, stride):
    if length <= tile_size:
        return [0]
    starts = list(range(0, max(1, length - tile_size + 1), max(1, stride)))
    if starts[-1] != length - tile_size:
        starts.append(length - tile_size)
    return starts


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


def mp_concat_list(tensors, dim=1):
    """mp_layers.py:65-86 with w=None (equal weights)."""
    n = len(tensors)
    w = 1.0 / n
    N = [t.shape[dim] for t in tensors]
    C = math.sqrt(sum(N) / (n * w * w))
    return torch.cat([t * (C / math.sqrt(t.shape[dim]) * w) for t in tensors], dim=dim)


def process_cond_img(cond_img, histogram_raw, cond_means, cond_stds, noise_level=0.0):
    """sample_diffusion_base.py:11-48 for NaN-free inputs. (B,7,4,4) -> (B,58)."""
    cond_img = cond_img.to(torch.float32)
    means = torch.as_tensor(cond_means, dtype=torch.float32).view(1, -1, 1, 1)
    stds = torch.as_tensor(cond_stds, dtype=torch.float32).view(1, -1, 1, 1)
    cond_img = (cond_img - means) / stds
    nl = (torch.as_tensor(noise_level, dtype=torch.float32) - 0.5) * np.sqrt(12)
    B = cond_img.shape[0]
    parts = [cond_img[:, 0:1].flatten(1), cond_img[:, 1:2].flatten(1),
             cond_img[:, 2:6, 1:3, 1:3].mean(dim=(2, 3)).flatten(1), cond_img[:, 6:7].flatten(1),
             torch.as_tensor(histogram_raw, dtype=torch.float32).view(-1, histogram_raw.shape[-1]).expand(B, -1),
             nl.view(-1, 1).expand(B, 1)]
    return mp_concat_list(parts, dim=1).float()


def synthetic_cond_grid(n_ty, n_tx, seed=0xC0DE):
    """SURVEY.md §8d synthetic conditioning: (1,7,n_ty+3,n_tx+3) standard normals."""
    return torch.from_numpy(rng.standard_normal(seed, (1, 7, n_ty + 3, n_tx + 3)))


def initial_noise_field(seed, H, W, channels=5, y0=0, x0=0):
    """Initial noise for a bounded canvas, cut from the absolute-coordinate field (tile 64)."""
    return torch.from_numpy(rng.gaussian_noise_patch(seed, y0, x0, H, W, channels=channels, tile_h=64, tile_w=64))[None]


@torch.no_grad()
def sample_base_diffusion_tiled(model, shape, cond_inputs, *, steps=20, tile_size=64, noise_seed=42 + 5819,
                                cond_means=None, cond_stds=None, histogram_raw=None, noise_level=0.0,
                                sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, rho=7.0,
                                tiles=None, return_parts=False, guide_model=None, guidance_scale=1.0, solver_order=2):
    """sample_diffusion_base.py:115-168 (B must be 1).  `tiles` optionally restricts to a subset of
    (ic, jc) tile indices (used for sharding tests); returns output/output_weights/sigma_data or parts."""
    B, C, H, W = shape
    assert B == 1
    stride = tile_size // 2
    sigmas, _ = schedule.karras_sigmas(steps, sigma_min, sigma_max, rho)
    orders = schedule.solver_orders(steps, solver_order=solver_order)
    weights = linear_weight_window(tile_size)[None, None]
    output = torch.zeros(shape)
    output_weights = torch.zeros(shape)
    initial_noise = initial_noise_field(noise_seed, H, W, C) * sigmas[0]
    h_starts, w_starts = tile_starts(H, tile_size, stride), tile_starts(W, tile_size, stride)
    cond_means = torch.zeros(7) if cond_means is None else cond_means
    cond_stds = torch.ones(7) if cond_stds is None else cond_stds
    histogram_raw = torch.zeros(1, 5) if histogram_raw is None else histogram_raw
    for ic, i0 in enumerate(h_starts):
        for jc, j0 in enumerate(w_starts):
            if tiles is not None and (ic, jc) not in tiles:
                continue
            if cond_inputs.ndim == 4:
                tile_cond = [process_cond_img(cond_inputs[..., ic:ic + 4, jc:jc + 4], histogram_raw, cond_means, cond_stds, noise_level)]
            else:
                tile_cond = [cond_inputs]
            x = initial_noise[..., i0:i0 + tile_size, j0:j0 + tile_size]
            m_prev = m_prev2 = None
            for i in range(steps):
                sigma = sigmas[i]
                xin = schedule.precondition_inputs(x, sigma, sigma_data)
                cn = schedule.trigflow_t(sigma.view(-1).expand(B), sigma_data)
                F_ = model(xin, cn, tile_cond)
                if guide_model is not None and guidance_scale != 1.0:   # autoguidance, sample_diffusion_base.py:155-160
                    F_g = guide_model(xin, cn, tile_cond)
                    F_ = F_g + guidance_scale * (F_ - F_g)
                x, m0_ = schedule.dpm_step(sigmas, i, orders[i], x, F_, m_prev, sigma_data, m_prev2=m_prev2)
                m_prev, m_prev2 = m0_, m_prev
            output[..., i0:i0 + tile_size, j0:j0 + tile_size] += x * weights
            output_weights[..., i0:i0 + tile_size, j0:j0 + tile_size] += weights
    if return_parts:
        return output, output_weights
    return output / output_weights / sigma_data


@torch.no_grad()
def sample_base_consistency_tiled(model, shape, cond_inputs, *, intermediate_t=0.0, tile_size=64,
                                  noise_seed=42 + 5819, cond_means=None, cond_stds=None, histogram_raw=None,
                                  noise_level=0.0, sigma_max=80.0, sigma_data=0.5):
    """sample_diffusion_base.py:171-268 — trig-flow consistency phases with a blend between phases.
    Phase k noise = portable field seeded noise_seed + k (world_pipeline.py:1153-1193 convention)."""
    B, C, H, W = shape
    assert B == 1
    stride = tile_size // 2
    sigma0 = torch.tensor(sigma_max, dtype=torch.float32)
    init_t = torch.atan(sigma0 / sigma_data)
    t_scalars = (init_t, torch.tensor(intermediate_t, dtype=torch.float32)) if intermediate_t > 0 else (init_t,)
    weights = linear_weight_window(tile_size)[None, None]
    h_starts, w_starts = tile_starts(H, tile_size, stride), tile_starts(W, tile_size, stride)
    cond_means = torch.zeros(7) if cond_means is None else cond_means
    cond_stds = torch.ones(7) if cond_stds is None else cond_stds
    histogram_raw = torch.zeros(1, 5) if histogram_raw is None else histogram_raw
    sample = torch.zeros(shape)
    for k, t_scalar in enumerate(t_scalars):
        step_noise = initial_noise_field(noise_seed + k, H, W, C)
        output = torch.zeros(shape)
        output_weights = torch.zeros(shape)
        for ic, i0 in enumerate(h_starts):
            for jc, j0 in enumerate(w_starts):
                if cond_inputs.ndim == 4:
                    tile_cond = [process_cond_img(cond_inputs[..., ic:ic + 4, jc:jc + 4], histogram_raw, cond_means, cond_stds, noise_level)]
                else:
                    tile_cond = [cond_inputs]
                z = step_noise[..., i0:i0 + tile_size, j0:j0 + tile_size] * sigma_data
                tile_sample = sample[..., i0:i0 + tile_size, j0:j0 + tile_size]
                t = t_scalar.view(1, 1, 1, 1).expand(B, 1, 1, 1)
                x_t = torch.cos(t) * tile_sample + torch.sin(t) * z
                pred = -model(x_t / sigma_data, t.flatten(), tile_cond)
                tile_samples = torch.cos(t) * x_t - torch.sin(t) * sigma_data * pred
                output[..., i0:i0 + tile_size, j0:j0 + tile_size] += tile_samples * weights
                output_weights[..., i0:i0 + tile_size, j0:j0 + tile_size] += weights
        sample = output / output_weights
    return sample / sigma_data


# ---- bounded decoder / coarse samplers (test infrastructure, CPU restatements) -------------------------------------------------------------
def _edm_tile(model, x, tile_cond_img, cond_list, steps, sigma_data=0.5, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """`steps` DPM-Solver++(2M) steps on one tile batch with conditioning-image channels concatenated after the sample channels
    (sample_diffusion_decoder.py:103-118, sample_coarse.py:104-117; a fresh step index per tile, see the note in the callers)."""
    sigmas, _ = schedule.karras_sigmas(steps, sigma_min, sigma_max, rho)
    orders = schedule.solver_orders(steps)
    m_prev = None
    for i in range(steps):
        sigma = sigmas[i]
        xin = torch.cat([schedule.precondition_inputs(x, sigma, sigma_data), tile_cond_img], dim=1)
        cn = schedule.trigflow_t(sigma.view(-1).expand(x.shape[0]), sigma_data)
        x, m_prev = schedule.dpm_step(sigmas, i, orders[i], x, model(xin, cn, cond_list), m_prev, sigma_data)
    return x


@torch.no_grad()
def sample_decoder_diffusion_tiled(model, cond_img, noise, tile_size=None, tile_stride=None, *, num_steps, sigma_data=0.5):
    """sample_diffusion_decoder.py:44-125 (no guidance, no score scaling).  The reference sets the timesteps once outside its tile loop and its
    scheduler's step index then runs off the table on the second tile (IndexError): every tile gets a fresh schedule here, which is what a
    single-tile call of the reference does."""
    b, c, h, w = noise.shape
    cond_img = cond_img.to(noise.dtype)
    if cond_img.shape[-2:] != (h, w):
        cond_img = torch.nn.functional.interpolate(cond_img, size=(h, w), mode="nearest")
    tile_size = min(h, w) if tile_size is None else tile_size
    tile_stride = tile_size if tile_stride is None else tile_stride
    weights = linear_weight_window(tile_size)[None, None]
    out, out_w = torch.zeros_like(noise), torch.zeros_like(noise)
    for i0 in tile_starts(h, tile_size, tile_stride):
        for j0 in tile_starts(w, tile_size, tile_stride):
            sl = (..., slice(i0, i0 + tile_size), slice(j0, j0 + tile_size))
            out[sl] += _edm_tile(model, noise[sl], cond_img[sl], [], num_steps, sigma_data) * weights
            out_w[sl] += weights
    return out / out_w


@torch.no_grad()
def sample_decoder_consistency_tiled(model, cond_img, noise, tile_size=None, tile_stride=None, *, intermediate_t=(), sigma_data=0.5, sigma_max=80.0):
    """sample_diffusion_decoder.py:129-211."""
    b, c, h, w = noise.shape
    cond_img = cond_img.to(noise.dtype)
    if cond_img.shape[-2:] != (h, w):
        cond_img = torch.nn.functional.interpolate(cond_img, size=(h, w), mode="nearest")
    tile_size = min(h, w) if tile_size is None else tile_size
    tile_stride = tile_size if tile_stride is None else tile_stride
    weights = linear_weight_window(tile_size)[None, None]
    out, out_w = torch.zeros_like(noise), torch.zeros_like(noise)
    ts = [torch.atan(torch.as_tensor(sigma_max / sigma_data, dtype=noise.dtype))] + [torch.tensor(t, dtype=noise.dtype) for t in intermediate_t]
    for i0 in tile_starts(h, tile_size, tile_stride):
        for j0 in tile_starts(w, tile_size, tile_stride):
            sl = (..., slice(i0, i0 + tile_size), slice(j0, j0 + tile_size))
            samples = torch.zeros((b, c, tile_size, tile_size), dtype=noise.dtype)
            for t_scalar in ts:
                t = t_scalar.view(1, 1, 1, 1).expand(b, 1, 1, 1)
                x_t = torch.cos(t) * samples + torch.sin(t) * (noise[sl] * sigma_data)
                pred = -model(torch.cat([x_t / sigma_data, cond_img[sl]], dim=1), t.flatten(), [])
                samples = torch.cos(t) * x_t - torch.sin(t) * sigma_data * pred
            out[sl] += samples * weights
            out_w[sl] += weights
    return out / out_w / sigma_data


@torch.no_grad()
def sample_coarse_tiled(model, cond_img, cond_snr, *, steps, cond_noise, init_noise, out_channels=6, tile_size=None, tile_stride=None, sigma_data=0.5):
    """sample_coarse.py:29-125 with the two torch.randn draws replaced by the given tensors (cond_noise like cond_img; init_noise: one
    (b, out_channels, T, T) per tile, row-major)."""
    b, c_cond, h, w = cond_img.shape
    tile_size = w if tile_size is None else tile_size
    tile_stride = tile_size if tile_stride is None else tile_stride
    weights = linear_weight_window(tile_size)[None, None]
    out = torch.zeros((b, out_channels, h, w))
    out_w = torch.zeros_like(out)
    snr = torch.as_tensor(cond_snr, dtype=torch.float32)
    t = torch.atan(snr)
    cond_list = [v.reshape(-1) for v in torch.log(torch.tan(t) / 8.0).transpose(0, 1)]
    cond_img = torch.cos(t).view(1, -1, 1, 1) * cond_img + torch.sin(t).view(1, -1, 1, 1) * cond_noise
    sigma0 = schedule.karras_sigmas(steps)[0][0]
    k = 0
    for i0 in tile_starts(h, tile_size, tile_stride):
        for j0 in tile_starts(w, tile_size, tile_stride):
            sl = (..., slice(i0, i0 + tile_size), slice(j0, j0 + tile_size))
            x = _edm_tile(model, init_noise[k] * sigma0, cond_img[sl], cond_list, steps, sigma_data) / sigma_data
            out[sl] += x * weights
            out_w[sl] += weights
            k += 1
    return out / out_w
