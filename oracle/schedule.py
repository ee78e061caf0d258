"""Oracle: cosine VP noise schedule and ancestral-step coefficients (test infrastructure).

Follows reference ``diffusion/noise_schedule.py:40-53`` (constants), ``:76-79`` (cosine
``marginal_log_mean_coeff``), ``:89-91`` (``marginal_prob``) and the per-step scalar
algebra of ``sampling.py:555-584,604-612``.  All arithmetic is torch fp32 on 0-d
tensors, in the reference's operation order.
"""
from __future__ import annotations

import math

import torch

COSINE_S = 0.008
COSINE_T = 0.9946  # noise_schedule.py:50
COSINE_LOG_ALPHA_0 = math.log(math.cos(COSINE_S / (1.0 + COSINE_S) * math.pi / 2.0))  # :46


def cosine_log_alpha(t: torch.Tensor) -> torch.Tensor:
    """noise_schedule.py:76-79."""
    return torch.log(torch.cos((t + COSINE_S) / (1.0 + COSINE_S) * math.pi / 2.0)) - COSINE_LOG_ALPHA_0


def marginal_prob(t: torch.Tensor):
    """noise_schedule.py:89-91 → (alpha_t, sigma_t)."""
    lm = cosine_log_alpha(t)
    return torch.exp(lm), torch.sqrt(1.0 - torch.exp(2.0 * lm))


def ancestral_coefficients(steps: int, eps: float = 1e-3, T: float = COSINE_T):
    """Per-step scalars of AncestralSampler (sampling.py:371,555-584,604-612).

    Returns dict of fp32 tensors [steps]: t, s, alpha_t, sigma_t, alpha_s, sigma_s,
    c_x (=alpha_{t|s} sigma_s^2 / sigma_t^2), c_pred (=alpha_s sigma^2_{t|s} / sigma_t^2),
    sigma (=sigma_{t|s} sigma_s / sigma_t), noise_level (=log(alpha_t^2 / sigma_t^2)).
    """
    t_array = torch.linspace(T, eps, steps)
    s_array = torch.cat([t_array[1:], torch.zeros(1)])
    out = {k: [] for k in ("t", "s", "alpha_t", "sigma_t", "alpha_s", "sigma_s", "c_x", "c_pred",
                           "sigma", "noise_level")}
    for i in range(steps):
        t, s = t_array[i], s_array[i]
        alpha_t, sigma_t = marginal_prob(t)
        alpha_s, sigma_s = marginal_prob(s)
        alpha_t_given_s = alpha_t / alpha_s
        sigma2_t_given_s = sigma_t ** 2 - alpha_t_given_s ** 2 * sigma_s ** 2
        sigma_t_given_s = torch.sqrt(sigma2_t_given_s)
        sigma = sigma_t_given_s * sigma_s / sigma_t
        vals = dict(t=t, s=s, alpha_t=alpha_t, sigma_t=sigma_t, alpha_s=alpha_s, sigma_s=sigma_s,
                    c_x=alpha_t_given_s * sigma_s ** 2 / sigma_t ** 2,
                    c_pred=alpha_s * sigma2_t_given_s / sigma_t ** 2,
                    sigma=sigma, noise_level=torch.log(alpha_t ** 2 / sigma_t ** 2))
        for k, v in vals.items():
            out[k].append(v.reshape(()))
    return {k: torch.stack(v) for k, v in out.items()}
