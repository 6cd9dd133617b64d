"""Panorama-demo plumbing (BASELINE configs[0]): the host-side pieces around a user-supplied denoiser.

Behavioural contract = annotated_infinite_panorama.py:57-102,145-150 (pinned bit-for-bit by tests/golden/geometry.npz); the
implementation is this package's own: a column-addressed noise strip with a small tile cache, a closed-form 1-D tent window, and a
bucketised phase split.  The SD-v1.5 U-Net / VAE of the demo live in `diffusers` (absent here: parity unpinned).
The noise stays on the host by design (numpy's PCG64DXSM ziggurat stream is the definition of the field)."""
from collections import OrderedDict

import numpy as np
import torch

_U32 = 0xFFFFFFFF


class NoiseStrip:
    """Infinite (channels, height, +-inf) Gaussian strip cut into `tile`-column blocks.  Block b is the float32 ziggurat stream of
    numpy's PCG64DXSM seeded with SeedSequence([seed, b mod 2^32]), so a column's values depend on (seed, column // tile) only and
    any two requests agree where they overlap.  Recently used blocks are kept (overlapping windows re-read the same block)."""

    def __init__(self, seed, channels=4, height=64, tile=256, keep=8):
        self.seed, self.shape, self.tile, self.keep = int(seed), (int(channels), int(height), int(tile)), int(tile), int(keep)
        self._blocks = OrderedDict()

    def block(self, b):
        hit = self._blocks.get(b)
        if hit is None:
            entropy = np.array([self.seed, b & _U32], dtype=np.uint32)
            gen = np.random.Generator(np.random.PCG64DXSM(np.random.SeedSequence(entropy)))
            hit = gen.standard_normal(self.shape, dtype=np.float32)
            self._blocks[b] = hit
            while len(self._blocks) > self.keep:
                self._blocks.popitem(last=False)
        else:
            self._blocks.move_to_end(b)
        return hit

    def columns(self, x0, width):
        """(channels, height, width) window starting at absolute column x0 (any sign)."""
        pieces, x, end = [], int(x0), int(x0) + int(width)
        while x < end:
            b, inside = divmod(x, self.tile)          # floor semantics: negative columns land in negative blocks
            take = min(self.tile - inside, end - x)
            pieces.append(self.block(b)[:, :, inside:inside + take])
            x += take
        return np.ascontiguousarray(np.concatenate(pieces, axis=2)) if len(pieces) != 1 else np.array(pieces[0], dtype=np.float32, order="C")


def tiled_gaussian_noise(seed, x0, width, channels=4, height=64, tile=256):
    """Function form with the demo's signature: one window of the strip identified by (seed, channels, height, tile)."""
    return NoiseStrip(seed, channels, height, tile, keep=2).columns(x0, width)


def linear_kernel(height, width):
    """Horizontal tent window, constant over rows: 1 at the centre column, 1e-3 at the two edge columns (fp32, same operation order
    as the demo so the blend weights are bit-identical)."""
    centre = (width - 1) / 2
    tent = torch.arange(width, dtype=torch.float32).sub_(centre).abs_().mul_(0.999).div_(centre).neg_().add_(1)
    return tent.repeat(height, 1)


def build_timestep_ranges(all_timesteps, thresholds):
    """Split a descending timestep vector into phases at the given thresholds: phase p holds the steps with exactly p thresholds
    strictly above them (so phase 0 is t >= max threshold, the last phase t < min threshold); empty phases are dropped."""
    cuts = sorted({int(t) for t in thresholds})
    if not cuts:
        return [all_timesteps]
    ts = torch.as_tensor(all_timesteps)
    # number of cuts <= t, counted from the top: phase = len(cuts) - #{c : c <= t}
    below_or_equal = torch.bucketize(ts, torch.tensor(cuts, dtype=ts.dtype), right=True)
    phase = len(cuts) - below_or_equal
    return [all_timesteps[phase == p] for p in range(len(cuts) + 1) if bool((phase == p).any())]


def normalize(weighted, clamp=1e-6):
    """(C+1,...) weighted sums -> (C,...) values; the weight channel is clamped from below (annotated_infinite_panorama.py:145-146)."""
    return weighted[:-1] / weighted[-1:].clamp(min=clamp)


def pack(values_chw, weight_hw):
    """values (C,H,W), weight (H,W) -> (C+1,H,W) = [values * weight, weight] (annotated_infinite_panorama.py:148-150)."""
    w = weight_hw[None]
    return torch.cat([values_chw * w, w], dim=0)


class DDIMSchedule:
    """The scheduler of the panorama demo (annotated_infinite_panorama.py:112-113: DDIMScheduler.from_config(SD-v1.5 config); set_timesteps(N)):
    1000 training steps, scaled-linear betas 0.00085 .. 0.012, 'leading' spacing with steps_offset 1, set_alpha_to_one False, epsilon prediction,
    eta 0.  Host scalars only; the update itself is td_ddim_cfg_step.  The arithmetic is diffusers' (absent here): restated from the published
    algorithm, parity unpinned (oracle/ddim.py says what is checked)."""
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps, self.steps_offset = int(num_train_timesteps), int(steps_offset)
        self.timesteps, self.num_inference_steps = None, None

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.num_train_timesteps // self.num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(self.num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)) + self.steps_offset
        return self

    def scale_model_input(self, sample, t=None):
        return sample

    def alphas(self, t):
        prev_t = int(t) - self.num_train_timesteps // self.num_inference_steps
        return float(self.alphas_cumprod[int(t)]), float(self.alphas_cumprod[prev_t] if prev_t >= 0 else self.alphas_cumprod[0])


def denoise(latent, timesteps, unet_fn, schedule, guidance_scale=7.5, engine=None):
    """annotated_infinite_panorama.py:125-134 on the engine: classifier-free-guided DDIM steps on a (1, C, H, W) device latent.
    unet_fn(inp (2, C, H, W), t) -> (2, C, H, W) is the caller's denoiser ([uncond, cond] halves; the demo's is SD-v1.5's UNet2DCondition, which is
    third-party and not part of this package); the guidance mix and the scheduler update are one HIP kernel per step (td_ddim_cfg_step)."""
    from ._lib import lib, check
    from .engine import get_engine, ptr
    latent = torch.as_tensor(latent, dtype=torch.float32)
    if not latent.is_cuda:
        latent = latent.cuda()
    latent = latent.contiguous().clone()
    eng = engine if engine is not None else get_engine(latent.device)
    n = latent.numel()
    for t in timesteps:
        inp = schedule.scale_model_input(torch.cat([latent] * 2), t)
        pred = torch.as_tensor(unet_fn(inp, t), dtype=torch.float32).to(latent.device).contiguous()
        a_t, a_prev = schedule.alphas(t)
        check(lib().td_ddim_cfg_step(eng._h, ptr(latent), ptr(pred[:1]), ptr(pred[1:]), n, float(guidance_scale), a_t, a_prev, ptr(latent)))
    return latent
