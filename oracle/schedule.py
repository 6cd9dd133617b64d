"""ORACLE: EDM Karras schedule + DPM-Solver++(2M) multistep update + trig-flow helpers (CPU, torch fp32).

Restates terrain_diffusion/scheduler/dpmsolver.py:285-342 (set_timesteps/_compute_karras_sigmas),
:226-258 (preconditioning), :454-561 (1st/2nd-order updates), :650-726 (step / order selection),
with the stateful float-equality step lookup (:618-648) replaced by an explicit step counter
(SURVEY.md Q8).  Scalars are fp32 torch 0-d tensors exactly as in the reference so that the
coefficient arithmetic rounds identically.
"""
import torch


def karras_sigmas(n, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """dpmsolver.py:329-342 with scaling_p=None; returns (sigmas[n+1] incl. final 0, timesteps[n])."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    sigmas = sigmas.to(torch.float32)
    timesteps = 0.25 * torch.log(sigmas)
    sigmas = torch.cat([sigmas, torch.tensor([0], dtype=torch.float32)])
    return sigmas, timesteps


def solver_orders(n, solver_order=2, lower_order_final=True, euler_at_final=False, final_zero=True):
    """dpmsolver.py:688-715 — which update each step uses (N=20 -> [1, 2x18, 1])."""
    orders, lower = [], 0
    for i in range(n):
        final = (i == n - 1) and (euler_at_final or (lower_order_final and n < 15) or final_zero)
        second = (i == n - 2) and lower_order_final and n < 15
        if solver_order == 1 or lower < 1 or final:
            orders.append(1)
        elif solver_order == 2 or lower < 2 or second:
            orders.append(2)
        else:
            orders.append(3)
        if lower < solver_order:
            lower += 1
    return orders


def precondition_inputs(sample, sigma, sigma_data=0.5):
    return sample * (1 / ((sigma ** 2 + sigma_data ** 2) ** 0.5))


def trigflow_t(sigma, sigma_data=0.5):
    return torch.atan(sigma / sigma_data)


def precondition_outputs(sample, model_output, sigma, sigma_data=0.5):
    """dpmsolver.py:245-258, prediction_type='epsilon'."""
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2) ** 0.5
    return c_skip * sample + c_out * model_output


def step_coefficients(sigmas, i, order):
    """fp32 scalars of one dpmsolver++ update at step i (alpha_t == 1), derived term-by-term from dpmsolver.py:481-482 / :529-540 / :586-613.
    Returns (a, b0, inv_r0, third) with
      order 1: x = a*x - b0*m0                                   a = s_t/s_s, b0 = exp(-h)-1
      order 2: x = a*x - b0*m0 - 0.5*b0*D1,                       D1 = inv_r0*(m0-m1)
      order 3: x = a*x - b0*D0 + c1*D1 - c2*D2                    third = (inv_r1, r0/(r0+r1), 1/(r0+r1), c1, c2), see dpm_step
    """
    s_t, s_s = sigmas[i + 1], sigmas[i]
    lam_t = torch.log(torch.tensor(1)) - torch.log(s_t)
    lam_s = torch.log(torch.tensor(1)) - torch.log(s_s)
    h = lam_t - lam_s
    a = s_t / s_s
    b0 = torch.exp(-h) - 1.0
    inv_r0, third = None, None
    if order >= 2:
        lam_s1 = torch.log(torch.tensor(1)) - torch.log(sigmas[i - 1])
        h0 = lam_s - lam_s1
        r0 = h0 / h
        inv_r0 = 1.0 / r0
        if order == 3:
            lam_s2 = torch.log(torch.tensor(1)) - torch.log(sigmas[i - 2])
            r1 = (lam_s1 - lam_s2) / h
            third = (1.0 / r1, r0 / (r0 + r1), 1.0 / (r0 + r1), (torch.exp(-h) - 1.0) / h + 1.0, (torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5)
    return a, b0, inv_r0, third


def dpm_step(sigmas, i, order, sample, model_output, m_prev, sigma_data=0.5, m_prev2=None):
    """One scheduler.step(): returns (prev_sample, x0_pred).  m_prev / m_prev2 = the x0 predictions of the previous two steps."""
    m0 = precondition_outputs(sample, model_output, sigmas[i], sigma_data)
    a, b0, inv_r0, third = step_coefficients(sigmas, i, order)
    if order == 1:
        x = a * sample - (1 * b0) * m0
    elif order == 2:
        d1 = inv_r0 * (m0 - m_prev)
        x = a * sample - (1 * b0) * m0 - 0.5 * (1 * b0) * d1
    else:   # dpmsolver.py:598-613
        inv_r1, f01, inv_r01, c1, c2 = third
        d1_0, d1_1 = inv_r0 * (m0 - m_prev), inv_r1 * (m_prev - m_prev2)
        d1 = d1_0 + f01 * (d1_0 - d1_1)
        d2 = inv_r01 * (d1_0 - d1_1)
        x = a * sample - (1 * b0) * m0 + (1 * c1) * d1 - (1 * c2) * d2
    return x, m0
