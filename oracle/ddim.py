"""TEST INFRASTRUCTURE -- CPU restatement of the DDIM update and the classifier-free-guidance mix of BASELINE configs[0]
(annotated_infinite_panorama.py:112-134: `DDIMScheduler.from_config(pipe.scheduler.config)`, `scale_model_input`, `step(...).prev_sample`,
`pred = uncond + GUIDANCE_SCALE * (cond - uncond)`).

PARITY UNPINNED: the arithmetic lives in the third-party `diffusers` package (requirements.txt: diffusers>=0.30.3), which is not installed in
the build image and is not vendored by the reference; no golden vector of the reference exists for it.  Restated from the published algorithm
(Song, Meng, Ermon 2021, "Denoising Diffusion Implicit Models", eq. 12 with eta = 0) as diffusers' DDIMScheduler runs it for the SD-v1.5
scheduler config: 1000 training steps, scaled-linear betas 0.00085 .. 0.012, epsilon prediction, timestep_spacing 'leading', steps_offset 1,
set_alpha_to_one False, clip_sample False.  `scale_model_input` is the identity and `init_noise_sigma` is 1 for DDIM.
What IS checked (tests/test_oracle_golden.py): the schedule's closed-form invariants and the exactness property of the deterministic update --
with the true noise as the model's prediction every step lands exactly on the forward-process sample of the next timestep.
"""
import numpy as np


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2   # "scaled_linear" in fp32 (torch.linspace)
    return np.cumprod(1.0 - betas, dtype=np.float32)


def timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    """'leading' spacing: (arange(N) * (T // N)).round()[::-1] + steps_offset"""
    ratio = num_train_timesteps // num_inference_steps
    return (np.arange(num_inference_steps) * ratio).round()[::-1].astype(np.int64) + steps_offset


def step_alphas(t, num_inference_steps, acp, num_train_timesteps=1000):
    """(alpha_prod_t, alpha_prod_t_prev) of the step that starts at timestep t; past the last step the 'final' alpha is alphas_cumprod[0]."""
    prev_t = int(t) - num_train_timesteps // num_inference_steps
    return float(acp[int(t)]), float(acp[prev_t] if prev_t >= 0 else acp[0])


def cfg_mix(uncond, cond, guidance_scale):
    return uncond + guidance_scale * (cond - uncond)


def ddim_step(latent, eps, a_t, a_prev):
    """eta = 0: x0 = (x - sqrt(1 - a_t) eps) / sqrt(a_t);  x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps   (float32 like the fp32 pipeline)"""
    latent, eps = np.asarray(latent, np.float32), np.asarray(eps, np.float32)
    x0 = (latent - np.float32((1.0 - a_t) ** 0.5) * eps) / np.float32(a_t ** 0.5)
    return np.float32(a_prev ** 0.5) * x0 + np.float32((1.0 - a_prev) ** 0.5) * eps


def denoise(latent, ts, num_inference_steps, unet_fn, guidance_scale=7.5, acp=None):
    """annotated_infinite_panorama.py:125-134 with unet_fn(inp (2, C, H, W), t) -> (2, C, H, W) standing in for pipe.unet(...).sample"""
    acp = alphas_cumprod() if acp is None else acp
    for t in ts:
        pred = unet_fn(np.concatenate([latent] * 2), int(t))
        a_t, a_prev = step_alphas(t, num_inference_steps, acp)
        latent = ddim_step(latent, cfg_mix(pred[:1], pred[1:], np.float32(guidance_scale)), a_t, a_prev)
    return latent
