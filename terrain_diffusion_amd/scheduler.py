"""EDMDPMSolverMultistepScheduler with the reference's surface (terrain_diffusion/scheduler/dpmsolver.py).

Host-side schedule arithmetic only (a few dozen fp32 scalars per run): the per-pixel update itself runs in
the HIP `dpm_step_kernel` when sampling goes through `sampling.sample_base_diffusion`.  `step()` is kept for
drop-in use by callers that drive the loop themselves (world_pipeline.py:941-949) and operates on torch tensors.
"""
from types import SimpleNamespace

import torch


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class EDMDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, sigma_schedule="karras", num_train_timesteps=1000,
                 prediction_type="epsilon", rho=7.0, solver_order=2, algorithm_type="dpmsolver++", solver_type="midpoint",
                 lower_order_final=True, euler_at_final=False, final_sigmas_type="zero", scaling_p=None, scaling_t=0.05, **_ignored):
        if sigma_schedule != "karras" or algorithm_type != "dpmsolver++" or solver_type != "midpoint" or scaling_p is not None:
            raise NotImplementedError("accelerated path covers the released configuration: karras / dpmsolver++ / midpoint")
        if solver_order not in (1, 2, 3) or prediction_type != "epsilon" or final_sigmas_type != "zero":
            raise NotImplementedError("solver_order in {1,2,3}, prediction_type='epsilon', final_sigmas_type='zero'")
        self.config = SimpleNamespace(sigma_min=sigma_min, sigma_max=sigma_max, sigma_data=sigma_data, rho=rho, solver_order=solver_order,
                                      lower_order_final=lower_order_final, euler_at_final=euler_at_final, final_sigmas_type=final_sigmas_type,
                                      prediction_type=prediction_type, num_train_timesteps=num_train_timesteps)
        self.set_timesteps(num_train_timesteps)
        self.num_inference_steps = None

    # dpmsolver.py:329-342
    def _compute_karras_sigmas(self, ramp):
        rho = self.config.rho
        mn, mx = self.config.sigma_min ** (1 / rho), self.config.sigma_max ** (1 / rho)
        return (mx + ramp * (mn - mx)) ** rho

    # dpmsolver.py:285-326
    def set_timesteps(self, num_inference_steps=None, device=None):
        self.num_inference_steps = num_inference_steps
        sigmas = self._compute_karras_sigmas(torch.linspace(0, 1, num_inference_steps)).to(torch.float32)
        self.timesteps = self.precondition_noise(sigmas)
        self.sigmas = torch.cat([sigmas, torch.tensor([0], dtype=torch.float32)])
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    @property
    def init_noise_sigma(self):
        return (self.config.sigma_max ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def precondition_inputs(self, sample, sigma):
        return sample * (1 / ((sigma ** 2 + self.config.sigma_data ** 2) ** 0.5))

    def precondition_noise(self, sigma):
        if not isinstance(sigma, torch.Tensor):
            sigma = torch.tensor([sigma])
        return 0.25 * torch.log(sigma)

    def trigflow_precondition_noise(self, sigma):
        return torch.atan(sigma / self.config.sigma_data)

    def precondition_outputs(self, sample, model_output, sigma):
        sd = self.config.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5
        return c_skip * sample + c_out * model_output

    # dpmsolver.py:650-726 with an explicit counter instead of the float-equality lookup (:618-648)
    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            idx = (self.timesteps == torch.as_tensor(timestep).to(self.timesteps.device)).nonzero()
            self._step_index = len(self.timesteps) - 1 if len(idx) == 0 else (idx[1].item() if len(idx) > 1 else idx[0].item())
        i, n = self._step_index, len(self.timesteps)
        sig = self.sigmas.to(sample.device)
        final = i == n - 1
        m0 = self.precondition_outputs(sample, model_output, sig[i])
        hist = [m for m in self.model_outputs if m is not None]
        m1 = hist[-1] if hist else None
        m2 = hist[-2] if len(hist) > 1 else None
        self.model_outputs = ([None] * self.config.solver_order + hist + [m0])[-self.config.solver_order:]
        a = sig[i + 1] / sig[i]
        h = -torch.log(sig[i + 1]) + torch.log(sig[i])
        b0 = torch.exp(-h) - 1.0
        second = i == n - 2 and self.config.lower_order_final and n < 15          # dpmsolver.py:694-696
        if self.config.solver_order == 1 or self.lower_order_nums < 1 or final:
            prev = a * sample - b0 * m0
        elif self.config.solver_order == 2 or self.lower_order_nums < 2 or second:
            h0 = -torch.log(sig[i]) + torch.log(sig[i - 1])
            d1 = (1.0 / (h0 / h)) * (m0 - m1)
            prev = a * sample - b0 * m0 - 0.5 * b0 * d1
        else:                                                                       # third-order multistep update, dpmsolver.py:563-615
            h0, h1 = -torch.log(sig[i]) + torch.log(sig[i - 1]), -torch.log(sig[i - 1]) + torch.log(sig[i - 2])
            r0, r1 = h0 / h, h1 / h
            d1_0, d1_1 = (1.0 / r0) * (m0 - m1), (1.0 / r1) * (m1 - m2)
            d1 = d1_0 + (r0 / (r0 + r1)) * (d1_0 - d1_1)
            d2 = (1.0 / (r0 + r1)) * (d1_0 - d1_1)
            prev = a * sample - b0 * m0 + ((torch.exp(-h) - 1.0) / h + 1.0) * d1 - ((torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5) * d2
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return SchedulerOutput(prev) if return_dict else (prev,)

    def __len__(self):
        return self.config.num_train_timesteps
