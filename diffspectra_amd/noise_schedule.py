"""VP noise schedule with the reference's interface (``diffusion/noise_schedule.py:6-122``).

Only the continuous ``cosine`` (the shipped configs, ``configs/diffspectra_qm9s.py:40``) and ``linear``
branches are provided; the scalar algebra is host-side torch fp32 in the reference's operation order so the
per-step coefficient table matches its CPU path bit for bit (tests/test_oracle_golden.py G1).
"""
from __future__ import annotations

import math

import torch


class NoiseScheduleVP:
    def __init__(self, schedule="cosine", betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20.0, dtype=torch.float32):
        if schedule not in ("linear", "cosine"):
            raise ValueError("Unsupported noise schedule {}: the MI355X path provides 'linear' and 'cosine' "
                             "(the discrete schedules of the reference are outside the sampling hot path)".format(schedule))
        self.schedule = schedule
        self.total_N = 1000
        self.beta_0, self.beta_1 = continuous_beta_0, continuous_beta_1
        self.cosine_s = 0.008
        self.cosine_beta_max = 999.0
        self.cosine_t_max = math.atan(self.cosine_beta_max * (1.0 + self.cosine_s) / math.pi) * 2.0 * (
            1.0 + self.cosine_s) / math.pi - self.cosine_s
        self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1.0 + self.cosine_s) * math.pi / 2.0))
        self.T = 0.9946 if schedule == "cosine" else 1.0

    def marginal_log_mean_coeff(self, t):
        if self.schedule == "linear":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.log(torch.cos((t + self.cosine_s) / (1.0 + self.cosine_s) * math.pi / 2.0)) - self.cosine_log_alpha_0

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_prob(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return torch.exp(lm), torch.sqrt(1.0 - torch.exp(2.0 * lm))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))

    def get_noiseLevel(self, t):
        alpha_t, sigma_t = self.marginal_alpha(t), self.marginal_std(t)
        return torch.log(alpha_t ** 2 / sigma_t ** 2)
