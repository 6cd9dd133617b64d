"""TEST INFRASTRUCTURE (oracle): CPU restatement of the per-window glue of the coarse and decoder stages of WorldPipeline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  Every function cites the reference
lines it follows (terrain_diffusion/inference/world_pipeline.py = "wp.py").  Pinned by tests/golden/stage_glue.npz, which
tests/golden/make_golden.py produced by running the reference's own `_coarse_inference`, `_pool_coarse_conditioning` and
`_decoder_inference` method bodies (AST-extracted, stub `self`) on seeded inputs.
"""
import numpy as np
import torch

from . import rng, schedule, tiling

SIGMA_DATA = 0.5


def pool_channel(x, n, mode):
    """wp.py:997-1005 — (1,H,W) -> (1,H/n,W/n) with max / min / avg pooling."""
    x = x.unsqueeze(0)
    if mode == "max":
        return torch.nn.functional.max_pool2d(x, kernel_size=n, stride=n).squeeze(0)
    if mode == "min":
        return -torch.nn.functional.max_pool2d(-x, kernel_size=n, stride=n).squeeze(0)
    return torch.nn.functional.avg_pool2d(x, kernel_size=n, stride=n).squeeze(0)


def pool_coarse_conditioning(img, n, elev_mode="avg", p5_mode="avg"):
    """wp.py:1007-1015 — channel 0 and 1 pooled with their own modes, the rest averaged."""
    if n == 1:
        return img
    rest = torch.nn.functional.avg_pool2d(img[2:].unsqueeze(0), kernel_size=n, stride=n).squeeze(0)
    return torch.cat([pool_channel(img[0:1], n, elev_mode), pool_channel(img[1:2], n, p5_mode), rest], dim=0)


def coarse_cond_inputs(cond_snr):
    """wp.py:976-978 — t_cond = atan(cond_snr); the five scalar conditional inputs are log(tan(t_cond)/8)."""
    t_cond = torch.atan(torch.as_tensor(cond_snr, dtype=torch.float32))
    vals = torch.log(torch.tan(t_cond) / 8.0)
    return t_cond, [v.view(-1) for v in vals]


def coarse_inference(model, ctx, *, seed, cond_map_fn, means, stds, cond_snr, pool_size=1, elev_mode="avg", p5_mode="avg", steps=20):
    """wp.py:909-959.  model: OracleUnet (coarse config); cond_map_fn(i1,i2,j1,j2) -> (5,64,64) synthetic map (wp.py:926).
    Returns the packed (7, 64/pool, 64/pool) window."""
    T, S = 64, 64 - 16
    means, stds = torch.as_tensor(means, dtype=torch.float32), torch.as_tensor(stds, dtype=torch.float32)
    _, i, j = ctx
    i1, j1 = i * (S // pool_size) * pool_size, j * (S // pool_size) * pool_size
    t_cond, cond_inputs = coarse_cond_inputs(cond_snr)
    smap = torch.as_tensor(cond_map_fn(i1, i1 + T, j1, j1 + T), dtype=torch.float32)
    sel = [0, 2, 3, 4, 5]
    smap = ((smap - means[sel, None, None]) / stds[sel, None, None])[None]
    cnoise = torch.from_numpy(rng.gaussian_noise_patch(seed, i1, j1, T, T, 5, T, T))[None]
    cond_img = torch.cos(t_cond.view(1, -1, 1, 1)) * smap + torch.sin(t_cond.view(1, -1, 1, 1)) * cnoise
    sig, orders = schedule.karras_sigmas(steps)[0], schedule.solver_orders(steps)
    x = torch.from_numpy(rng.gaussian_noise_patch(seed + 1, i1, j1, T, T, 6, T, T))[None] * sig[0]
    m_prev = None
    with torch.no_grad():
        for k in range(steps):
            xin = torch.cat([schedule.precondition_inputs(x, sig[k]), cond_img], dim=1)
            F_ = model(xin, schedule.trigflow_t(sig[k].view(-1)), cond_inputs)
            x, m_prev = schedule.dpm_step(sig, k, orders[k], x, F_, m_prev)
    x = x / SIGMA_DATA
    x = x * stds.view(1, -1, 1, 1) + means.view(1, -1, 1, 1)
    x[0, 1] = x[0, 0] - x[0, 1]
    out = x[0] if pool_size == 1 else pool_coarse_conditioning(x[0], pool_size, elev_mode, p5_mode)
    w = tiling.linear_weight_window(T // pool_size)
    return torch.cat([out * w[None], w[None]], dim=0)


def decoder_inference(model, ctx, latents, *, seed, tile_size=512, tile_stride=384, lc=8, t_list=None):
    """wp.py:1209-1242.  latents: the (6, tile/lc, tile/lc) un-normalised window of the latent stage; returns (2, tile, tile)."""
    T = tile_size
    if t_list is None:
        t_list = [torch.atan(schedule.karras_sigmas(20)[0][0] / SIGMA_DATA)]   # wp.py:1252 (scheduler.sigmas[0] == sigma_max)
    lat = (latents[:-1] / latents[-1:])[:4].view(1, 4, T // lc, T // lc)
    up = torch.nn.functional.interpolate(lat, size=(T, T), mode="nearest")
    sample = torch.zeros(1, 1, T, T)
    with torch.no_grad():
        for i, t in enumerate(t_list):
            t = torch.as_tensor(t, dtype=torch.float32).view(1, 1, 1, 1)
            z = torch.from_numpy(rng.gaussian_noise_patch(seed + 5819 + i, ctx[1] * tile_stride, ctx[2] * tile_stride, T, T, 1, T, T))[None] * SIGMA_DATA
            x_t = torch.cos(t) * sample + torch.sin(t) * z
            pred = -model(torch.cat([x_t / SIGMA_DATA, up], dim=1), t.view(1), [])
            sample = torch.cos(t) * x_t - torch.sin(t) * SIGMA_DATA * pred
    sample = sample / SIGMA_DATA
    w = tiling.linear_weight_window(T)
    return torch.cat([sample[0] * w[None], w[None]], dim=0)


def synthetic_coarse_map(i1, i2, j1, j2):
    """deterministic stand-in for WorldPipeline._conditioning_model_input (the synthetic-map generator is out of scope):
    five smooth fields of the absolute cell coordinates."""
    y = torch.arange(i1, i2, dtype=torch.float32)[:, None]
    x = torch.arange(j1, j2, dtype=torch.float32)[None, :]
    return torch.stack([torch.sin(0.05 * y + 0.3 * k) * torch.cos(0.03 * x - 0.2 * k) * (1.0 + 0.5 * k) + 0.1 * k for k in range(5)])


def process_latent_conditioning(cond_img, histogram_raw, cond_means, cond_stds, noise_level, *, seed, seed_offset=0):
    """wp.py:1018-1050, literally.  NOTE the reference indexes the BATCH dimension when it fills NaNs (`cond_img[0:1]`, `cond_img[1:2]`):
    every NaN of sample 0 (all channels, after normalisation) becomes cond_means[0], of sample 1 cond_means[1]; further samples keep
    their NaNs and the NaN climate means are then drawn from the portable RNG seeded seed + 9999 + seed_offset.  (B,7,4,4) -> (B,58)."""
    cond_means = torch.as_tensor(cond_means, dtype=torch.float32)
    cond_stds = torch.as_tensor(cond_stds, dtype=torch.float32)
    cond_img = (cond_img.to(torch.float32) - cond_means.view(1, -1, 1, 1)) / cond_stds.view(1, -1, 1, 1)
    cond_img[0:1] = cond_img[0:1].nan_to_num(float(cond_means[0]))
    cond_img[1:2] = cond_img[1:2].nan_to_num(float(cond_means[1]))
    clim = cond_img[:, 2:6, 1:3, 1:3].mean(dim=(2, 3))
    nan_mask = torch.isnan(clim)
    cnt = int(nan_mask.sum())
    if cnt > 0:
        clim[nan_mask] = torch.from_numpy(rng.standard_normal((seed + 9999 + seed_offset), (cnt,)))
    nl = (torch.as_tensor(noise_level, dtype=torch.float32) - 0.5) * np.sqrt(12)
    parts = [cond_img[:, 0:1].flatten(1), cond_img[:, 1:2].flatten(1), clim.flatten(1), cond_img[:, 6:7].flatten(1),
             torch.as_tensor(histogram_raw, dtype=torch.float32), nl.view(-1, 1)]
    return tiling.mp_concat_list(parts, dim=1).float()


def latent_inference(model, ctxs, samples, cond_windows, t, *, seed, seed_offset, histogram_raw, cond_means, cond_stds):
    """wp.py:1052-1131.  samples: None or packed (6,64,64) previous-phase sums; cond_windows: packed (7,4,4) coarse windows.
    Returns the packed (6,64,64) windows."""
    T, S = 64, 32
    w = tiling.linear_weight_window(T)
    t = torch.as_tensor(t, dtype=torch.float32)
    tv = t.view(1, 1, 1, 1)
    outs = []
    with torch.no_grad():
        for k, ctx in enumerate(ctxs):
            if samples is None or samples[k] is None:
                sample = torch.zeros(1, 5, T, T)
            else:
                s_ = torch.as_tensor(samples[k], dtype=torch.float32)
                sample = (s_[:-1] / s_[-1:] * SIGMA_DATA)[None]
            c = torch.as_tensor(cond_windows[k], dtype=torch.float32)
            cimg = torch.cat([c[:-1] / c[-1:], torch.ones(1, 4, 4)], dim=0)[None]
            cond = process_latent_conditioning(cimg, histogram_raw, cond_means, cond_stds, torch.tensor(0.0), seed=seed, seed_offset=ctx[1] * 65536 + ctx[2])
            z = torch.from_numpy(rng.gaussian_noise_patch(seed + seed_offset, ctx[1] * S, ctx[2] * S, T, T, 5, T, T))[None] * SIGMA_DATA
            x_t = torch.cos(tv) * sample + torch.sin(tv) * z
            pred = -model(x_t / SIGMA_DATA, t.view(1), [cond])
            out = (torch.cos(tv) * x_t - torch.sin(tv) * SIGMA_DATA * pred) / SIGMA_DATA
            outs.append(torch.cat([out[0] * w[None], w[None]], dim=0))
    return outs
