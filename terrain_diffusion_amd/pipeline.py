"""InfiniteDiffusion stages on the engine: the lazy, unbounded counterparts of the bounded samplers (coarse -> latent -> decoder).

Mirrors the structure of WorldPipeline._build_latent_stage / _latent_inference (terrain_diffusion/inference/world_pipeline.py:
1052-1131, 1133-1203): a chain of InfiniteTensors, one per trig-flow phase, windows of `tile` stride `tile//2`, every window output
packed as (C+1, tile, tile) = (sample * w, w); a phase reads the blended (summed) previous phase through `args_windows`, divides by
the weight channel and re-noises with the portable field seeded `seed + 5819 + phase` at absolute coordinates.  All missing
windows of a request are batched through the U-Net (`batch_size`), noise / U-Net / consistency update run in the HIP engine.
The latent stage's conditioning source is a callback (the glue between the stages -- Laplacian / climate composition,
`_process_latent_conditioning`, the synthetic-map generator -- is out of scope).

build_coarse_stage / build_decoder_stage mirror `_build_coarse_stage` + `_coarse_inference` (world_pipeline.py:909-992) and
`_build_decoder_stage` + `_decoder_inference` (world_pipeline.py:1209-1270): same window geometry, seeds, channel arithmetic and
packing; the 20-step solver loop / trig-flow step, the noise field and the U-Net run in the HIP engine, batched over windows.
"""
import math

import torch

from ._lib import lib, check
from .engine import ptr
from .infinite_tensor import InfiniteTensor, MemoryTileStore, TensorWindow
from . import noise as _noise
from .sampling import _linear_weight_window


def _stacked(x):
    """argument slices of a batch of windows: the device-resident graph hands them over as ONE (n, ...) tensor (DeviceWindowTensor.gather_many),
    the host-resident graph as a list of per-window tensors"""
    return x.to(torch.float32) if torch.is_tensor(x) else torch.stack([torch.as_tensor(p, dtype=torch.float32) for p in x])


LATENT_COND_MEAN = (14.99, 11.65, 15.87, 619.26, 833.12, 69.40, 0.66)     # world_pipeline.py:1137-1138
LATENT_COND_STD = (21.72, 21.78, 10.40, 452.29, 738.09, 34.59, 0.47)


def build_latent_stage(model, sigma_data=0.5, sigma_max=80.0, *, seed, cond_fn=None, coarse=None, histogram_raw=None,
                       cond_means=LATENT_COND_MEAN, cond_stds=LATENT_COND_STD, intermediate_ts=(math.atan(0.35 / 0.5),), T=2,
                       onestep_latent=False, tile=64, channels=5, batch_size=16, tile_store=None, tensor_prefix="latents", device_resident=False):
    """Returns the final-phase InfiniteTensor of shape (channels+1, None, None)  (world_pipeline.py:1133-1203).

    Conditioning source, one of
      coarse=<InfiniteTensor (7,None,None)> (+ histogram_raw): the reference's wiring -- each latent window reads the (7,4,4) coarse
        window at the same index through TensorWindow(size=(7,4,4), stride=(7,1,1), offset=(0,-1,-1)), normalises it, appends the
        ones mask and goes through process_latent_conditioning with seed_offset = i*65536 + j (world_pipeline.py:1080-1088);
      cond_fn(ctxs) -> (len(ctxs), cond_dim): a caller-supplied conditioning matrix for window indices ctxs = [(0, i, j), ...].
    T=2: one InfiniteTensor per trig-flow phase, blended between phases; T=1: all phases inside one window function
    (world_pipeline.py:1149-1172).  onestep_latent stops after the first phase.
    device_resident=True: windows stay in HBM (DeviceWindowTensor / DeviceTileStore), regions are assembled by the engine's blend kernel and
    slices are device tensors; False: host tensors and a host tile store exactly as the reference keeps them (world_pipeline.py:1129)."""
    from .sampling import process_latent_conditioning_windows
    if (cond_fn is None) == (coarse is None):
        raise ValueError("give exactly one of cond_fn= or coarse=")
    stride = tile // 2
    store = tile_store if tile_store is not None else MemoryTileStore()
    dev = model.device
    w = _linear_weight_window(tile, dev)[0, 0].cpu()
    win = TensorWindow(size=(channels + 1, tile, tile), stride=(channels + 1, stride, stride))
    cwin = TensorWindow(size=(7, 4, 4), stride=(7, 1, 1), offset=(0, -1, -1))
    t_init = float(torch.atan(torch.tensor(sigma_max, dtype=torch.float32) / sigma_data))
    ts = (t_init,) + (() if onestep_latent else tuple(float(torch.as_tensor(t, dtype=torch.float32)) for t in intermediate_ts))
    hist = None if histogram_raw is None else torch.as_tensor(histogram_raw, dtype=torch.float32).view(1, -1)

    def conditioning(ctxs, conds):
        if coarse is None:
            return torch.as_tensor(cond_fn(ctxs), dtype=torch.float32)
        # all windows of the batch at once (round 4; the per-window loop cost ~30 tiny launches and one host synchronisation per window)
        c = _stacked(conds)                                                                           # (n, 7, 4, 4) packed coarse slices
        cimg = torch.cat([c[:, :-1] / c[:, -1:], torch.ones(c.shape[0], 1, 4, 4, device=c.device)], dim=1)   # world_pipeline.py:1080-1083
        return process_latent_conditioning_windows(cimg, hist, cond_means, cond_stds, 0.0)

    def infer_raw(phase, t, ctxs, prevs, conds):
        """one trig-flow phase on a batch of windows -> (n, channels, tile, tile) on the device, divided by sigma_data (world_pipeline.py:1129)"""
        n = len(ctxs)
        origins = [(c[1] * stride, c[2] * stride) for c in ctxs]
        z = _noise.gaussian_noise_patches(seed + 5819 + phase, origins, tile, tile, channels=channels, tile_h=tile, tile_w=tile, device=dev)
        sample = None
        if prevs is not None:  # (C+1, tile, tile) un-normalised sums -> sample * sigma_data (world_pipeline.py:1078)
            pv = _stacked(prevs).to(dev)
            sample = ((pv[:, :-1] / pv[:, -1:]) * sigma_data).contiguous()
        cond = conditioning(ctxs, conds).to(dev).contiguous()
        out = torch.empty_like(z)
        check(lib().td_sample_consistency(model._h, n, tile, tile, float(t), float(sigma_data), ptr(sample), ptr(z), ptr(cond), ptr(out)))
        return out / sigma_data

    def infer(phase, t, ctxs, prevs, conds):
        out = infer_raw(phase, t, ctxs, prevs, conds).cpu()  # world_pipeline.py:1129: the reference keeps window outputs on the host
        return [torch.cat([o * w[None], w[None]], dim=0) for o in out]

    if device_resident:
        from .infinite_tensor import DeviceWindowTensor
        cw = cwin if coarse is not None else None
        dsrc, dwin = ((coarse,), (cw,)) if coarse is not None else ((), ())
        mk = lambda f, args, wins, tid: DeviceWindowTensor(channels, f, tile, stride, model.engine, args=args, args_windows=wins, tile_store=tile_store,
                                                            tensor_id=tid, batch_size=batch_size)
        if T == 1:
            w_dev = w.to(dev)

            def f_t1d(ctxs, conds=None):
                outs = None
                for k, t in enumerate(ts):   # later phases read the window's own previous output (no blend): pack it the way a slice would arrive
                    prevs = None if outs is None else torch.cat([outs * w_dev[None, None], w_dev[None, None].expand(outs.shape[0], 1, -1, -1)], dim=1)
                    outs = infer_raw(k, t, ctxs, prevs, conds)
                return outs
            lat = mk(f_t1d, dsrc, dwin, f"{tensor_prefix}_T1")
        else:
            lat = mk(lambda ctxs, conds=None: infer_raw(0, ts[0], ctxs, None, conds), dsrc, dwin, f"{tensor_prefix}_phase0")
            for k, t in enumerate(ts[1:], 1):
                lat = mk(lambda ctxs, prevs, conds=None, k=k, t=t: infer_raw(k, t, ctxs, prevs, conds), (lat,) + dsrc, (win,) + dwin, f"{tensor_prefix}_phase{k}")
        lat.infer = infer
        return lat

    src_args, src_wins = ((coarse,), (cwin,)) if coarse is not None else ((), ())
    shape = (channels + 1, None, None)
    if T == 1:
        def f_t1(ctxs, conds=None):
            outs = None
            for k, t in enumerate(ts):
                outs = infer(k, t, ctxs, outs, conds)
            return outs
        lat = InfiniteTensor(shape, f_t1, win, args=src_args, args_windows=src_wins, tile_store=store, tensor_id=f"{tensor_prefix}_T1",
                             batch_size=batch_size)
        lat.infer = infer
        return lat
    lat = InfiniteTensor(shape, lambda ctxs, conds=None: infer(0, ts[0], ctxs, None, conds), win, args=src_args, args_windows=src_wins,
                         tile_store=store, tensor_id=f"{tensor_prefix}_phase0", batch_size=batch_size)
    for k, t in enumerate(ts[1:], 1):
        lat = InfiniteTensor(shape, lambda ctxs, prevs, conds=None, k=k, t=t: infer(k, t, ctxs, prevs, conds), win, args=(lat,) + src_args,
                             args_windows=(win,) + src_wins, tile_store=store, tensor_id=f"{tensor_prefix}_phase{k}", batch_size=batch_size)
    lat.infer = infer   # the per-window arithmetic, exposed for parity tests
    return lat


def _pool_channel(x, n, mode):
    """world_pipeline.py:997-1005"""
    x = x.unsqueeze(0)
    if mode == "max":
        return torch.nn.functional.max_pool2d(x, kernel_size=n, stride=n).squeeze(0)
    if mode == "min":
        return -torch.nn.functional.max_pool2d(-x, kernel_size=n, stride=n).squeeze(0)
    return torch.nn.functional.avg_pool2d(x, kernel_size=n, stride=n).squeeze(0)


def pool_coarse_conditioning(img, n, elev_mode="avg", p5_mode="avg"):
    """world_pipeline.py:1007-1015 -- (C, H*n, W*n) -> (C, H, W): channels 0 / 1 with their own pooling modes, the rest averaged."""
    if n == 1:
        return img
    rest = torch.nn.functional.avg_pool2d(img[2:].unsqueeze(0), kernel_size=n, stride=n).squeeze(0)
    return torch.cat([_pool_channel(img[0:1], n, elev_mode), _pool_channel(img[1:2], n, p5_mode), rest], dim=0)


def build_coarse_stage(model, scheduler, *, seed, cond_map_fn, coarse_means, coarse_stds, cond_snr, coarse_pooling=1,
                       elev_coarse_pool_mode="avg", p5_coarse_pool_mode="avg", steps=20, batch_size=16, tile_store=None,
                       tensor_id="base_coarse_map", device_resident=False):
    """InfiniteTensor (7, None, None) of packed coarse windows (world_pipeline.py:961-992).
    model: coarse EDMUnet2D (11 -> 6 channels, five "float" conditional inputs); cond_map_fn(i1, i2, j1, j2) -> (5, 64, 64) is the
    reference's `_conditioning_model_input` (synthetic map, channels [0,2,3,4,5] of the coarse statistics)."""
    from .sampling import sample_tiles_edm
    T, S = 64, 64 - 16
    pool = int(coarse_pooling)
    if T % pool or S % pool:
        raise ValueError(f"coarse_pooling {pool} must divide the tile size {T} and stride {S}")
    dev = model.device
    means, stds = torch.as_tensor(coarse_means, dtype=torch.float32), torch.as_tensor(coarse_stds, dtype=torch.float32)
    sel = [0, 2, 3, 4, 5]
    sigma_data = float(scheduler.config.sigma_data)
    w = _linear_weight_window(T // pool, dev)[0, 0].cpu()
    t_cond = torch.atan(torch.as_tensor(cond_snr, dtype=torch.float32))                     # world_pipeline.py:976-978
    cond_vals = torch.log(torch.tan(t_cond) / 8.0)
    tc = t_cond.view(1, -1, 1, 1).to(dev)

    def f_raw(ctxs):
        """(n, 6, T/pool, T/pool) de-normalised, pooled coarse windows on the device"""
        n = len(ctxs)
        origins = [(c[1] * (S // pool) * pool, c[2] * (S // pool) * pool) for c in ctxs]
        m5, s5 = means[sel, None, None].to(dev), stds[sel, None, None].to(dev)
        smap = torch.stack([(torch.as_tensor(cond_map_fn(i1, i1 + T, j1, j1 + T), dtype=torch.float32).to(dev) - m5) / s5 for i1, j1 in origins])
        cnoise = _noise.gaussian_noise_patches(seed, origins, T, T, channels=5, tile_h=T, tile_w=T, device=dev)
        cond_img = (torch.cos(tc) * smap + torch.sin(tc) * cnoise).contiguous()
        scheduler.set_timesteps(steps)
        x = _noise.gaussian_noise_patches(seed + 1, origins, T, T, channels=6, tile_h=T, tile_w=T, device=dev) * float(scheduler.sigmas[0])
        cond = model.cond_rows([v.view(1).expand(n) for v in cond_vals], n, dev)
        sample_tiles_edm(model, scheduler, x, cond, steps, cond_img=cond_img)
        x = x.float() / sigma_data                                                          # world_pipeline.py:950-952
        x = x * stds.view(1, -1, 1, 1).to(dev) + means.view(1, -1, 1, 1).to(dev)
        x[:, 1] = x[:, 0] - x[:, 1]
        return torch.stack([pool_coarse_conditioning(x[k], pool, elev_coarse_pool_mode, p5_coarse_pool_mode) for k in range(n)])

    def f(ctxs):
        outs = f_raw(ctxs).cpu()                                                            # the reference's `.cpu().float()` (:951)
        return [torch.cat([o * w[None], w[None]], dim=0) for o in outs]

    if device_resident:
        from .infinite_tensor import DeviceWindowTensor
        return DeviceWindowTensor(6, f_raw, T // pool, S // pool, model.engine, tile_store=tile_store, tensor_id=tensor_id, batch_size=batch_size)
    win = TensorWindow(size=(7, T // pool, T // pool), stride=(7, S // pool, S // pool))
    return InfiniteTensor((7, None, None), f, win, tile_store=tile_store if tile_store is not None else MemoryTileStore(), tensor_id=tensor_id,
                          batch_size=batch_size)


def build_decoder_stage(model, latents, *, seed, sigma_data=0.5, sigma_max=80.0, tile_size=512, tile_stride=384, latent_compression=8,
                        extra_ts=(), batch_size=4, tile_store=None, tensor_id="init_residual_map", device_resident=False):
    """InfiniteTensor (2, None, None) of packed decoder windows (world_pipeline.py:1244-1270) over the latent stage's tensor `latents`
    ((6, None, None): 5 latent channels + weight).  One trig-flow step at t = atan(sigma_max / sigma_data) (+ `extra_ts`)."""
    from .sampling import consistency_step
    T, S, lc = int(tile_size), int(tile_stride), int(latent_compression)
    dev = model.device
    w = _linear_weight_window(T, dev)[0, 0].cpu()
    t_list = (float(torch.atan(torch.tensor(sigma_max, dtype=torch.float32) / sigma_data)),) + tuple(float(torch.as_tensor(t, dtype=torch.float32)) for t in extra_ts)

    def f_raw(ctxs, lats):
        n = len(ctxs)
        origins = [(c[1] * S, c[2] * S) for c in ctxs]
        lv = _stacked(lats).to(dev)
        lat = (lv[:, :-1] / lv[:, -1:])[:, :4]                                                                                     # :1223
        up = lat.repeat_interleave(T // lat.shape[-2], dim=-2).repeat_interleave(T // lat.shape[-1], dim=-1).contiguous()         # nearest, :1224
        sample = None
        for i, t in enumerate(t_list):
            z = _noise.gaussian_noise_patches(seed + 5819 + i, origins, T, T, channels=1, tile_h=T, tile_w=T, device=dev)
            sample = consistency_step(model, t, sigma_data, sample, z, cond=None, cond_img=up)
        return sample.float() / sigma_data

    def f(ctxs, lats):
        out = f_raw(ctxs, lats).cpu()                                                       # world_pipeline.py:1241
        return [torch.cat([o * w[None], w[None]], dim=0) for o in out]

    owin = TensorWindow(size=(2, T, T), stride=(2, S, S))
    iwin = TensorWindow(size=(6, T // lc, T // lc), stride=(6, S // lc, S // lc))
    if device_resident:
        from .infinite_tensor import DeviceWindowTensor
        return DeviceWindowTensor(1, f_raw, T, S, model.engine, args=(latents,), args_windows=(iwin,), tile_store=tile_store, tensor_id=tensor_id,
                                  batch_size=batch_size)
    return InfiniteTensor((2, None, None), f, owin, args=(latents,), args_windows=(iwin,),
                          tile_store=tile_store if tile_store is not None else MemoryTileStore(), tensor_id=tensor_id, batch_size=batch_size)
