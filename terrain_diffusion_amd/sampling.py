"""Bounded tiled samplers with the reference's call surface, executed by the HIP engine.

Mirrors terrain_diffusion/training/evaluation/sample_diffusion_base.py:51-168 (sample_base_diffusion) and
:171-268 (sample_base_consistency); helper geometry from training/evaluation/__init__.py:3-22.  Differences
by design: all tiles of a phase are batched through the engine (they are independent), the initial noise is cut
from the portable absolute-coordinate field (world_pipeline.py:66-115) instead of torch.randn, and the blend is a
deterministic gather in the reference's loop order.
"""
import ctypes as C
import math

import numpy as np
import torch

from ._lib import lib, check
from .engine import ptr
from . import noise as _noise


from .geometry import tile_starts as _tile_starts  # noqa: E402  (reference name kept as an alias)


def _linear_weight_window(size, device="cuda", dtype=torch.float32):
    from .engine import get_engine
    eng = get_engine(device)
    out = torch.empty((size, size), dtype=torch.float32, device=torch.device("cuda", eng.device_id))
    check(lib().td_linear_weight_window(eng._h, size, ptr(out)))
    return out[None, None]


def _process_cond_img(cond_img, histogram_raw, cond_means, cond_stds, noise_level=0.0):
    """sample_diffusion_base.py:11-48 for NaN-free conditioning (host, (B,7,4,4) -> (B,58)); tiny, stays on the host."""
    cond_img = torch.as_tensor(cond_img, dtype=torch.float32).cpu()
    means = torch.as_tensor(cond_means, dtype=torch.float32).view(1, -1, 1, 1)
    stds = torch.as_tensor(cond_stds, dtype=torch.float32).view(1, -1, 1, 1)
    cond_img = (cond_img - means) / stds
    # NaN fill exactly as the reference writes it (sample_diffusion_base.py:36-46, batch-dimension indexing and all):
    # sample 0's NaNs become cond_means[0], sample 1's cond_means[1]; NaN climate means are drawn from torch's global generator there,
    # which no other implementation can reproduce -> here they come from the portable stream (seed 9999 + count), documented deviation
    cm = torch.as_tensor(cond_means, dtype=torch.float32).flatten()
    cond_img[0:1] = torch.nan_to_num(cond_img[0:1], nan=float(cm[0]))
    cond_img[1:2] = torch.nan_to_num(cond_img[1:2], nan=float(cm[1]))
    B = cond_img.shape[0]
    nl = (torch.as_tensor(noise_level, dtype=torch.float32) - 0.5) * np.sqrt(12)
    hist = torch.as_tensor(histogram_raw, dtype=torch.float32)
    clim = cond_img[:, 2:6, 1:3, 1:3].mean(dim=(2, 3))
    nan_mask = torch.isnan(clim)
    if bool(nan_mask.any()):
        from .noise import standard_normal
        cnt = int(nan_mask.sum())
        clim[nan_mask] = torch.from_numpy(standard_normal(9999 + cnt, (cnt,), dtype=np.float32))
    parts = [cond_img[:, 0:1].flatten(1), cond_img[:, 1:2].flatten(1), clim.flatten(1),
             cond_img[:, 6:7].flatten(1), hist.view(-1, hist.shape[-1]).expand(B, -1), nl.view(-1, 1).expand(B, 1)]
    n = len(parts)
    Cc = math.sqrt(sum(p.shape[1] for p in parts) / (n * (1.0 / n) ** 2))
    return torch.cat([p * (Cc / math.sqrt(p.shape[1]) * (1.0 / n)) for p in parts], dim=1).float()


def process_latent_conditioning(cond_img, histogram_raw, cond_means, cond_stds, noise_level=0.0, *, seed, seed_offset=0):
    """WorldPipeline._process_latent_conditioning (world_pipeline.py:1018-1050): (B,7,4,4) -> (B,58), on whatever device cond_img lives
    (the device-resident pipeline hands over HBM tensors; 58 floats per window never need the host).
    Keeps the reference's behaviour to the letter, including its batch-dimension NaN fill: every NaN of sample 0 becomes cond_means[0]
    and of sample 1 cond_means[1] (after normalisation); NaN climate means of further samples are drawn from the portable RNG seeded
    seed + 9999 + seed_offset (the engine's generator: same stream as portable_rng.standard_normal)."""
    from .noise import standard_normal
    cond_img = torch.as_tensor(cond_img, dtype=torch.float32)
    dev = cond_img.device
    cond_means = torch.as_tensor(cond_means, dtype=torch.float32)
    cond_stds = torch.as_tensor(cond_stds, dtype=torch.float32)
    cond_img = (cond_img - cond_means.view(1, -1, 1, 1).to(dev)) / cond_stds.view(1, -1, 1, 1).to(dev)
    cond_img[0:1] = cond_img[0:1].nan_to_num(float(cond_means[0]))
    cond_img[1:2] = cond_img[1:2].nan_to_num(float(cond_means[1]))
    clim = cond_img[:, 2:6, 1:3, 1:3].mean(dim=(2, 3))
    nan_mask = torch.isnan(clim)
    cnt = int(nan_mask.sum())
    if cnt > 0:
        clim[nan_mask] = torch.from_numpy(standard_normal(seed + 9999 + seed_offset, (cnt,), dtype=np.float32)).to(dev)
    B = cond_img.shape[0]
    nl = ((torch.as_tensor(noise_level, dtype=torch.float32) - 0.5) * np.sqrt(12)).to(dev)
    hist = torch.as_tensor(histogram_raw, dtype=torch.float32).to(dev)
    parts = [cond_img[:, 0:1].flatten(1), cond_img[:, 1:2].flatten(1), clim.flatten(1), cond_img[:, 6:7].flatten(1),
             hist.view(-1, hist.shape[-1]).expand(B, -1), nl.view(-1, 1).expand(B, 1)]
    n = len(parts)
    Cc = math.sqrt(sum(p.shape[1] for p in parts) / (n * (1.0 / n) ** 2))
    return torch.cat([p * (Cc / math.sqrt(p.shape[1]) * (1.0 / n)) for p in parts], dim=1).float()


_DEV_CONST = {}


def _dev_const(t, dev):
    """small constant (statistics vectors, the histogram) as a device tensor, uploaded ONCE per (content, device): `.to(device)` of a pageable
    host tensor waits for the copy, which would put a host synchronisation into every batch of an otherwise enqueue-only stage"""
    t = torch.as_tensor(t, dtype=torch.float32)
    if t.device == dev or t.is_cuda:
        return t.to(dev)
    key = (t.numpy().tobytes(), tuple(t.shape), str(dev))
    v = _DEV_CONST.get(key)
    if v is None:
        if len(_DEV_CONST) > 64:
            _DEV_CONST.clear()
        v = _DEV_CONST[key] = t.to(dev)
    return v


def process_latent_conditioning_windows(cond_imgs, histogram_raw, cond_means, cond_stds, noise_level=0.0):
    """process_latent_conditioning applied to n windows ONE AT A TIME, as the reference's latent stage calls it (world_pipeline.py:1080-1088: a
    (1,7,4,4) image per window), evaluated for all windows at once: (n,7,4,4) -> (n,58) with a dozen tensor ops instead of ~30 per window.
    With a batch of one the reference's batch-dimension NaN fill (`cond_img[0:1].nan_to_num(cond_means[0])`) covers every channel of the
    window, so no NaN reaches the climate means and the per-window RNG fill (seed + 9999 + seed_offset) is never drawn: the vectorised form
    needs neither a seed nor a device-to-host synchronisation (the per-window form reads `nan_mask.sum()` back for every window)."""
    x = torch.as_tensor(cond_imgs, dtype=torch.float32)
    dev = x.device
    cond_means = torch.as_tensor(cond_means, dtype=torch.float32)
    cond_stds = torch.as_tensor(cond_stds, dtype=torch.float32)
    x = ((x - _dev_const(cond_means.view(1, -1, 1, 1), dev)) / _dev_const(cond_stds.view(1, -1, 1, 1), dev)).nan_to_num(float(cond_means[0]))
    clim = x[:, 2:6, 1:3, 1:3].mean(dim=(2, 3))
    B = x.shape[0]
    nl = (torch.as_tensor(noise_level, dtype=torch.float32) - 0.5) * np.sqrt(12)
    nl = torch.full((1, 1), float(nl), dtype=torch.float32, device=dev) if nl.numel() == 1 and not nl.is_cuda else nl.to(dev)   # scalar: a fill, not an upload
    hist = _dev_const(histogram_raw, dev)
    parts = [x[:, 0:1].flatten(1), x[:, 1:2].flatten(1), clim.flatten(1), x[:, 6:7].flatten(1), hist.view(-1, hist.shape[-1]).expand(B, -1), nl.view(-1, 1).expand(B, 1)]
    n = len(parts)
    Cc = math.sqrt(sum(p.shape[1] for p in parts) / (n * (1.0 / n) ** 2))
    return torch.cat([p * (Cc / math.sqrt(p.shape[1]) * (1.0 / n)) for p in parts], dim=1).float()


def _tile_conditioning(cond_inputs, tiles, histogram_raw, cond_means, cond_stds, noise_level):
    if cond_inputs.ndim == 4:
        return torch.cat([_process_cond_img(cond_inputs[..., ic:ic + 4, jc:jc + 4], histogram_raw, cond_means, cond_stds, noise_level)
                          for ic, jc in tiles], dim=0)
    return torch.as_tensor(cond_inputs, dtype=torch.float32).view(1, -1).expand(len(tiles), -1).contiguous()


def blend_windows(engine, canvas, tiles, tile_idx, h_starts, w_starts, size, accumulate=True):
    """canvas (C+1,Hc,Wc) += windows (deterministic gather, reference loop order)."""
    C_, Hc, Wc = canvas.shape[0] - 1, canvas.shape[1], canvas.shape[2]
    rs = np.asarray(h_starts, dtype=np.int32)
    cs = np.asarray(w_starts, dtype=np.int32)
    wi = np.asarray([t[0] for t in tile_idx], dtype=np.int32)
    wj = np.asarray([t[1] for t in tile_idx], dtype=np.int32)
    check(lib().td_blend_windows(engine._h, ptr(canvas), C_, Hc, Wc, size, len(rs), C.c_void_p(rs.ctypes.data), len(cs), C.c_void_p(cs.ctypes.data),
                                 len(wi), C.c_void_p(wi.ctypes.data), C.c_void_p(wj.ctypes.data), ptr(tiles), int(accumulate)))
    return canvas


def blend_normalize(engine, canvas, scale=1.0):
    C_, Hc, Wc = canvas.shape[0] - 1, canvas.shape[1], canvas.shape[2]
    out = torch.empty((C_, Hc, Wc), dtype=torch.float32, device=canvas.device)
    check(lib().td_blend_normalize(engine._h, ptr(canvas), C_, Hc, Wc, float(scale), ptr(out)))
    return out


@torch.no_grad()
def sample_tiles_edm(model, scheduler, x, cond, steps, cond_img=None, guide_model=None, guidance_scale=1.0):
    """Runs `steps` DPM-Solver++ steps on a batch of independent tiles (device tensor x: [n,C,H,W], scaled noise). In place.
    cond_img: optional [n,Cc,H,W] conditioning-image channels concatenated after the sample channels in the model input
    (coarse stage: world_pipeline.py:946 `torch.cat([scaled_in, cond_img], dim=1)`)."""
    scheduler.set_timesteps(steps)
    sig = scheduler.sigmas.to(torch.float32).cpu().contiguous()
    model.engine.set_option("solver_order", int(getattr(scheduler.config, "solver_order", 2)))
    # dpmsolver.py:694-696: the second-to-last step of a third-order run drops to second order only when config.lower_order_final is set (and the
    # run has < 15 steps); the engine's order schedule follows the scheduler's flag instead of assuming the released default
    model.engine.set_option("lower_order_final", int(bool(getattr(scheduler.config, "lower_order_final", True))))
    n, _, H, W = x.shape
    cimg = 0 if cond_img is None else cond_img.shape[1]
    if guide_model is not None and guidance_scale != 1.0:   # autoguidance (sample_diffusion_base.py:105-110)
        if cond_img is not None:
            raise NotImplementedError("autoguidance with conditioning-image channels")
        check(lib().td_sample_edm_guided(model._h, guide_model._h, float(guidance_scale), n, H, W, steps, ptr(sig), float(scheduler.config.sigma_data), ptr(cond), ptr(x)))
        return x
    check(lib().td_sample_edm_img(model._h, n, H, W, steps, ptr(sig), float(scheduler.config.sigma_data), ptr(cond), ptr(cond_img), cimg, ptr(x)))
    return x


@torch.no_grad()
def consistency_step(model, t, sigma_data, sample, z, cond=None, cond_img=None):
    """One trig-flow consistency step on a batch of tiles (world_pipeline.py:1097-1129 latent stage, :1229-1239 decoder):
    x_t = cos t * sample + sin t * sigma_d * z ; out = cos t * x_t + sin t * sigma_d * model([x_t/sigma_d | cond_img], t, cond).
    sample may be None (zeros).  Returns a new device tensor shaped like z."""
    n, _, H, W = z.shape
    out = torch.empty_like(z)
    cimg = 0 if cond_img is None else cond_img.shape[1]
    check(lib().td_sample_consistency_img(model._h, n, H, W, float(t), float(sigma_data), ptr(sample), ptr(z), ptr(cond), ptr(cond_img), cimg, ptr(out)))
    return out


@torch.no_grad()
def sample_base_diffusion(model, scheduler, shape, cond_inputs, *, cond_means, cond_stds, noise_level=0.0, histogram_raw, dtype=torch.float32,
                          steps=15, guide_model=None, guidance_scale=1.0, generator=None, tile_size=None, weight_window_fn=None,
                          noise_seed=42 + 5819, noise_origin=(0, 0), tiles=None, max_batch=64, return_canvas=False, return_windows=False):
    """Reference signature + (noise_seed, noise_origin, tiles, max_batch, return_canvas, return_windows).
    `tiles`: optional subset of (ic, jc) window indices to run (multi-GPU sharding); `return_canvas` returns the
    un-normalised (C+1,H,W) accumulator instead of output/weights/sigma_data; `return_windows` additionally returns the
    pre-blend window outputs [(ic, jc)] -> (C, T, T) device tensors (parity tests look at windows before the blend mixes them)."""
    if weight_window_fn is not None:
        raise NotImplementedError("custom weight windows")
    B, C_, H, W = shape
    if B != 1:
        raise NotImplementedError("B == 1 (one canvas) per call")
    eng, dev = model.engine, model.device
    sd = float(scheduler.config.sigma_data)
    scheduler.set_timesteps(steps)
    sigma0 = float(scheduler.sigmas[0])
    if tile_size is None:
        tile_size_eff, h_starts, w_starts = None, [0], [0]
        th, tw = H, W
    else:
        stride = tile_size // 2
        h_starts, w_starts = _tile_starts(H, tile_size, stride), _tile_starts(W, tile_size, stride)
        th = tw = tile_size
    cond_inputs = torch.as_tensor(cond_inputs, dtype=torch.float32)
    if tile_size is not None and cond_inputs.ndim == 1 and len(h_starts) * len(w_starts) > 1:
        raise ValueError(f"cond_inputs must be a tensor image for tiled sampling. Cond inputs must have width {len(w_starts)+3} and height {len(h_starts)+3}.")
    if cond_inputs.ndim == 4:
        assert cond_inputs.shape[-1] == len(w_starts) + 3 and cond_inputs.shape[-2] == len(h_starts) + 3
    all_tiles = [(ic, jc) for ic in range(len(h_starts)) for jc in range(len(w_starts))]
    run = all_tiles if tiles is None else [t for t in all_tiles if t in set(tiles)]
    canvas = torch.zeros((C_ + 1, H, W), dtype=torch.float32, device=dev)
    windows = {}
    for b0 in range(0, len(run), max_batch):
        chunk = run[b0:b0 + max_batch]
        origins = [(noise_origin[0] + h_starts[ic], noise_origin[1] + w_starts[jc]) for ic, jc in chunk]
        # initial_noise[..., i0:i1, j0:j1] of one shared field (sample_diffusion_base.py:124,145) == windows of the absolute field
        # the bounded sampler's noise field is this package's own convention (the reference draws torch.randn here): 64x64 noise tiles,
        # or one tile size that holds the window when the window is larger
        nth, ntw = max(64, th), max(64, tw)
        x = _noise.gaussian_noise_patches(noise_seed, origins, th, tw, channels=C_, tile_h=nth, tile_w=ntw, scale=sigma0, device=dev)
        cond = _tile_conditioning(cond_inputs, chunk, histogram_raw, cond_means, cond_stds, noise_level).to(dev).contiguous()
        sample_tiles_edm(model, scheduler, x, cond, steps, guide_model=guide_model, guidance_scale=guidance_scale)
        if tile_size is None:
            return x
        if return_windows:
            windows.update({t: x[k].clone() for k, t in enumerate(chunk)})
        blend_windows(eng, canvas, x, chunk, h_starts, w_starts, tile_size, accumulate=True)
    out = canvas if return_canvas else blend_normalize(eng, canvas, 1.0 / sd)[None]
    return (out, windows) if return_windows else out


@torch.no_grad()
def sample_independent_tiles(model, scheduler, origins, cond, *, steps=20, tile_size=64, channels=5, noise_seed=42 + 5819, return_raw=False):
    """A batch of INDEPENDENT single-tile jobs (BASELINE configs[1], batched the way the reference batches latent tiles with
    `latents_batch_size`, world_pipeline.py:292,326-330): tile i = sample_base_diffusion(shape=(1,C,T,T), tile_size=T) at noise
    origin origins[i] with conditioning vector cond[i] (n,58).  Every stage runs in the engine: noise field -> 20 x (U-Net +
    DPM-Solver++ step) -> pack with the linear window / normalise / divide by sigma_data.  Returns (n, C, T, T) on the device."""
    eng, dev = model.engine, model.device
    sd = float(scheduler.config.sigma_data)
    scheduler.set_timesteps(steps)
    n = len(origins)
    x = _noise.gaussian_noise_patches(noise_seed, origins, tile_size, tile_size, channels=channels, tile_h=64, tile_w=64,
                                      scale=float(scheduler.sigmas[0]), device=dev)
    cond = torch.as_tensor(cond, dtype=torch.float32).to(dev).contiguous()
    sample_tiles_edm(model, scheduler, x, cond, steps)
    if return_raw:
        return x
    # n single-window canvases stacked vertically: (C+1, n*T, T); same pack/normalise arithmetic as one canvas per tile
    canvas = torch.empty((channels + 1, n * tile_size, tile_size), dtype=torch.float32, device=dev)
    blend_windows(eng, canvas, x, [(i, 0) for i in range(n)], [i * tile_size for i in range(n)], [0], tile_size, accumulate=False)
    out = blend_normalize(eng, canvas, 1.0 / sd)
    return out.view(channels, n, tile_size, tile_size).permute(1, 0, 2, 3)


@torch.no_grad()
def sample_base_consistency(model, scheduler, shape, cond_inputs, *, cond_means, cond_stds, noise_level=0.0, histogram_raw, intermediate_t=0.0,
                            dtype=torch.float32, generator=None, tile_size=None, weight_window_fn=None, noise=None,
                            noise_seed=42 + 5819, noise_origin=(0, 0), max_batch=64):
    """sample_diffusion_base.py:171-268: trig-flow consistency phases, blend between phases (the InfiniteDiffusion pattern)."""
    B, C_, H, W = shape
    if B != 1 or tile_size is None or weight_window_fn is not None:
        raise NotImplementedError
    eng, dev = model.engine, model.device
    sd = float(scheduler.config.sigma_data)
    init_t = math.atan(float(scheduler.config.sigma_max) / sd) if noise is None else None
    sigma0 = scheduler.sigmas[0] if noise is not None else None
    t0 = float(torch.atan(torch.tensor(scheduler.config.sigma_max, dtype=torch.float32) / sd))
    t_scalars = (t0, float(torch.tensor(intermediate_t, dtype=torch.float32))) if intermediate_t > 0 else (t0,)
    stride = tile_size // 2
    h_starts, w_starts = _tile_starts(H, tile_size, stride), _tile_starts(W, tile_size, stride)
    cond_inputs = torch.as_tensor(cond_inputs, dtype=torch.float32)
    tiles = [(ic, jc) for ic in range(len(h_starts)) for jc in range(len(w_starts))]
    origins = [(h_starts[ic], w_starts[jc]) for ic, jc in tiles]
    cond_all = _tile_conditioning(cond_inputs, tiles, histogram_raw, cond_means, cond_stds, noise_level).to(dev).contiguous()
    sample = None
    for k, t in enumerate(t_scalars):
        canvas = torch.zeros((C_ + 1, H, W), dtype=torch.float32, device=dev)
        for b0 in range(0, len(tiles), max_batch):
            sl = slice(b0, b0 + max_batch)
            if noise is None:
                z = _noise.gaussian_noise_patches(noise_seed + k, [(noise_origin[0] + y, noise_origin[1] + x) for y, x in origins[sl]], tile_size, tile_size,
                                                  channels=C_, tile_h=64, tile_w=64, device=dev)
            else:
                z = torch.stack([torch.as_tensor(noise[k])[0, :, y:y + tile_size, x:x + tile_size] for y, x in origins[sl]]).to(dev, torch.float32).contiguous()
            prev = None
            if sample is not None:
                prev = torch.stack([sample[:, y:y + tile_size, x:x + tile_size] for y, x in origins[sl]]).contiguous()
            out = torch.empty_like(z)
            cond = cond_all[sl].contiguous()
            check(lib().td_sample_consistency(model._h, z.shape[0], tile_size, tile_size, float(t), sd, ptr(prev), ptr(z), ptr(cond), ptr(out)))
            blend_windows(eng, canvas, out, tiles[sl], h_starts, w_starts, tile_size, accumulate=True)
        sample = blend_normalize(eng, canvas, 1.0)
    return (sample / sd)[None]


# ------------------------------------------------------------------------------------------------------------------------------------------
# Bounded twins of the decoder and coarse stages (training/evaluation/sample_diffusion_decoder.py:44-211, sample_coarse.py:29-125): the same
# call surface as the reference functions, every tile of every batch item run through the engine in batches (tiles are independent), blended
# with the engine's deterministic gather in the reference's loop order.
def _cond_tiles(cond_img, h, w, device):
    """cond_img -> (b, Cc, h, w) fp32 on the device, nearest-resized when its spatial size differs (sample_diffusion_decoder.py:82-84)."""
    cond_img = torch.as_tensor(cond_img).to(device=device, dtype=torch.float32)
    if tuple(cond_img.shape[-2:]) != (h, w):
        cond_img = torch.nn.functional.interpolate(cond_img, size=(h, w), mode="nearest")
    return cond_img.contiguous()


def _blend_batch(eng, tiles, b, n_tiles, tile_idx, h_starts, w_starts, tile_size, h, w, scale):
    """tiles (n_tiles*b, C, T, T) ordered tile-major, batch-minor -> (b, C, h, w): out / out_w of the reference loops, times `scale`."""
    C_ = tiles.shape[1]
    outs = []
    for k in range(b):
        canvas = torch.zeros((C_ + 1, h, w), dtype=torch.float32, device=tiles.device)
        blend_windows(eng, canvas, tiles[k::b].contiguous(), tile_idx, h_starts, w_starts, tile_size, accumulate=False)
        outs.append(blend_normalize(eng, canvas, scale))
    return torch.stack(outs)


def _tile_geometry(h, w, tile_size, tile_stride, default):
    tile_size = default if tile_size is None else int(tile_size)
    tile_stride = tile_size if tile_stride is None else int(tile_stride)
    h_starts, w_starts = _tile_starts(h, tile_size, tile_stride), _tile_starts(w, tile_size, tile_stride)
    tile_idx = [(ic, jc) for ic in range(len(h_starts)) for jc in range(len(w_starts))]
    return tile_size, h_starts, w_starts, tile_idx


@torch.no_grad()
def sample_decoder_diffusion_tiled(model, scheduler, cond_img, noise, tile_size=None, tile_stride=None, *, num_steps=None, guidance_model=None,
                                   guidance_scale=1.0, score_scaling=1.0, weight_window_fn=None, max_batch=64):
    """sample_diffusion_decoder.py:44-125: tiled conditional EDM sampling of a decoder model.  `noise` is the initial sample as the caller
    scaled it (the reference uses it as is), `cond_img` is concatenated after the sample channels.  Returns out / out_w (no sigma_data
    division -- the reference has none here).  Not supported (raise): a guide model together with conditioning-image channels, score scaling
    other than 1, custom weight windows."""
    if weight_window_fn is not None or score_scaling != 1.0:
        raise NotImplementedError("custom weight windows / score scaling")
    if guidance_model is not None and guidance_scale != 1.0:
        raise NotImplementedError("autoguidance with conditioning-image channels")
    if num_steps is not None:
        scheduler.set_timesteps(num_steps)
    steps = len(scheduler.timesteps)
    eng, dev = model.engine, model.device
    noise = torch.as_tensor(noise).to(device=dev, dtype=torch.float32)
    b, c, h, w = noise.shape
    cond_img = _cond_tiles(cond_img, h, w, dev)
    T, h_starts, w_starts, tile_idx = _tile_geometry(h, w, tile_size, tile_stride, min(h, w))
    outs = []
    jobs = [(i0, j0) for i0 in h_starts for j0 in w_starts]
    per = max(1, max_batch // b)
    for a in range(0, len(jobs), per):
        chunk = jobs[a:a + per]
        x = torch.cat([noise[:, :, i0:i0 + T, j0:j0 + T] for i0, j0 in chunk]).contiguous()       # tile-major, batch-minor
        ci = torch.cat([cond_img[:, :, i0:i0 + T, j0:j0 + T] for i0, j0 in chunk]).contiguous()
        outs.append(sample_tiles_edm(model, scheduler, x, None, steps, cond_img=ci))
    return _blend_batch(eng, torch.cat(outs), b, len(jobs), tile_idx, h_starts, w_starts, T, h, w, 1.0)


@torch.no_grad()
def sample_decoder_consistency_tiled(model, scheduler, cond_img, noise, tile_size=None, tile_stride=None, *, intermediate_t=None, weight_window_fn=None,
                                     max_batch=64):
    """sample_diffusion_decoder.py:129-211: n-step trig-flow consistency sampling of a decoder model per tile (every step re-noises with the
    SAME tile noise), blended, divided by sigma_data."""
    if weight_window_fn is not None:
        raise NotImplementedError("custom weight windows")
    eng, dev = model.engine, model.device
    noise = torch.as_tensor(noise).to(device=dev, dtype=torch.float32)
    b, c, h, w = noise.shape
    cond_img = _cond_tiles(cond_img, h, w, dev)
    sd = float(scheduler.config.sigma_data)
    init_t = float(torch.atan(torch.as_tensor(float(scheduler.sigmas[0]) / sd, dtype=torch.float32)))
    if intermediate_t is None:
        extra = []
    elif torch.is_tensor(intermediate_t):
        extra = [float(t) for t in intermediate_t.flatten().to(torch.float32)]
    elif isinstance(intermediate_t, (list, tuple)):
        extra = [float(torch.tensor(t, dtype=torch.float32)) for t in intermediate_t]
    else:
        extra = [float(torch.tensor(float(intermediate_t), dtype=torch.float32))]
    T, h_starts, w_starts, tile_idx = _tile_geometry(h, w, tile_size, tile_stride, min(h, w))
    jobs = [(i0, j0) for i0 in h_starts for j0 in w_starts]
    per = max(1, max_batch // b)
    outs = []
    for a in range(0, len(jobs), per):
        chunk = jobs[a:a + per]
        z = torch.cat([noise[:, :, i0:i0 + T, j0:j0 + T] for i0, j0 in chunk]).contiguous()
        ci = torch.cat([cond_img[:, :, i0:i0 + T, j0:j0 + T] for i0, j0 in chunk]).contiguous()
        sample = None
        for t in [init_t] + extra:
            sample = consistency_step(model, t, sd, sample, z, cond=None, cond_img=ci)
        outs.append(sample)
    return _blend_batch(eng, torch.cat(outs), b, len(jobs), tile_idx, h_starts, w_starts, T, h, w, 1.0 / sd)


@torch.no_grad()
def sample_coarse_tiled(model, scheduler, cond_img, cond_snr, *, steps=15, tile_size=None, tile_stride=None, weight_window_fn=None, generator=None,
                        dtype=torch.float32, cond_noise=None, init_noise=None, noise_seed=42, max_batch=64):
    """sample_coarse.py:29-125: the coarse model on a bounded (b, C_cond, h, w) conditioning image.  The conditioning image is mixed with
    noise at the per-channel SNR, every tile starts from sigma_0 * noise, runs `steps` DPM-Solver++ steps with the five log(tan(atan(snr)) / 8)
    conditional scalars, is divided by sigma_data and blended.
    The reference draws both noises from torch's generators (torch.randn_like / torch.randn), which nothing else can reproduce: pass
    `cond_noise` (b, C_cond, h, w) and `init_noise` (list of (b, C_out, T, T), one per tile in row-major tile order) to pin them, otherwise
    they come from the portable stream seeded `noise_seed` (+1 + tile index for the tiles)."""
    if weight_window_fn is not None:
        raise NotImplementedError("custom weight windows")
    eng, dev = model.engine, model.device
    cond_img = torch.as_tensor(cond_img).to(device=dev, dtype=torch.float32)
    assert cond_img.ndim == 4, "cond_img must be [B, C, H, W]"
    b, c_cond, h, w = cond_img.shape
    T, h_starts, w_starts, tile_idx = _tile_geometry(h, w, tile_size, tile_stride, w)
    if h < T or w < T:   # the reference sizes each tile from its cond slice (sample_coarse.py:88-92); this sampler draws square T x T tiles
        raise ValueError(f"sample_coarse_tiled: the conditioning image ({h} x {w}) is smaller than the tile ({T} x {T}); pass tile_size <= min(h, w)")
    c_out = int(model.config["out_channels"])
    snr = torch.as_tensor(cond_snr, dtype=torch.float32)
    t_cond = torch.atan(snr)
    cond_vals = torch.log(torch.tan(t_cond) / 8.0)                                        # sample_coarse.py:7-26
    cond_inputs = [v.reshape(-1) for v in cond_vals.reshape(-1, cond_vals.shape[-1]).transpose(0, 1)]
    tc = t_cond.reshape(1, -1, 1, 1).to(dev)
    if cond_noise is None:
        cond_noise = torch.from_numpy(_noise.standard_normal(noise_seed, tuple(cond_img.shape)))
    cond_img = (torch.cos(tc) * cond_img + torch.sin(tc) * torch.as_tensor(cond_noise).to(dev, torch.float32)).contiguous()
    scheduler.set_timesteps(int(steps))
    sd = float(scheduler.config.sigma_data)
    sigma0 = float(scheduler.sigmas[0])
    jobs = [(i0, j0) for i0 in h_starts for j0 in w_starts]
    per = max(1, max_batch // b)
    outs = []
    for a in range(0, len(jobs), per):
        chunk = list(enumerate(jobs))[a:a + per]
        xs = []
        for k, (i0, j0) in chunk:
            z = init_noise[k] if init_noise is not None else torch.from_numpy(_noise.standard_normal(noise_seed + 1 + k, (b, c_out, T, T)))
            xs.append(torch.as_tensor(z).to(dev, torch.float32) * sigma0)
        x = torch.cat(xs).contiguous()
        ci = torch.cat([cond_img[:, :, i0:i0 + T, j0:j0 + T] for _, (i0, j0) in chunk]).contiguous()
        n = x.shape[0]
        cond = model.cond_rows([v if v.numel() == 1 else v.repeat(len(chunk)) for v in cond_inputs], n, dev)   # tile-major, batch-minor rows
        sample_tiles_edm(model, scheduler, x, cond, int(steps), cond_img=ci)
        outs.append(x)
    return _blend_batch(eng, torch.cat(outs), b, len(jobs), tile_idx, h_starts, w_starts, T, h, w, 1.0 / sd)
