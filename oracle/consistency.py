"""CPU restatement of the consistency-model forward and sampler (test infrastructure only).

Reference lines restated (/root/reference/tqdne/consistency_model.py):
  * forward (skip/out scalings, raw sigma as timestep) ... consistency_model.py:63-79
  * sample (1 NFE + one per refinement sigma) ............ consistency_model.py:81-106
The refinement noise of line 103 is *uniform* (``rand_like``); it is injected here.
"""

from __future__ import annotations

from typing import Optional, Sequence

import torch

Tensor = torch.Tensor


def _bcast(v: Tensor, ndim: int) -> Tensor:
    return v[(...,) + (None,) * (ndim - v.ndim)]


def forward(net, sample: Tensor, sigma: Tensor, cond_sample=None, cond=None,
            sigma_min: float = 0.002, sigma_data: float = 0.5) -> Tensor:
    x_in = sample if cond_sample is None else torch.cat((sample, cond_sample), dim=1)
    cs = sigma_data**2 / ((sigma - sigma_min) ** 2 + sigma_data**2)
    co = (sigma_data * (sigma - sigma_min)) / (sigma_data**2 + sigma**2) ** 0.5
    out = net(x_in, sigma, cond)
    return _bcast(co, sample.dim()) * out + _bcast(cs, sample.dim()) * sample


def sample(net, start_unit_noise: Tensor, sigmas: Sequence[float] = (),
           refine_uniform_noises: Sequence[Tensor] = (), cond_sample=None, cond=None,
           sigma_min: float = 0.002, sigma_max: float = 80.0, sigma_data: float = 0.5) -> Tensor:
    ones = torch.ones(start_unit_noise.shape[0])
    x = forward(net, start_unit_noise, ones * sigma_max, cond_sample, cond, sigma_min, sigma_data)
    for s, u in zip(sigmas, refine_uniform_noises):
        x = x + u * s
        x = forward(net, x, ones * s, cond_sample, cond, sigma_min, sigma_data)
    return x


def ict_schedule(global_step: int, max_steps: int, initial_timesteps: int = 10, final_timesteps: int = 1280,
                 sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0) -> Tensor:
    """consistency_model.py:121-138: discretisation that doubles over training; returns the sigma grid (num_timesteps,)."""
    import numpy as np
    prime = np.floor(max_steps / (np.log2(np.floor(final_timesteps / initial_timesteps)) + 1))
    num = initial_timesteps * 2 ** np.floor(global_step / prime)
    num = min(num, final_timesteps) + 1
    rho_inv = 1.0 / rho
    steps = torch.arange(num) / (num - 1)
    sig = sigma_min**rho_inv + steps * (sigma_max**rho_inv - sigma_min**rho_inv)
    return sig**rho


def ict_timestep_pdf(sigmas: Tensor, lognormal_mean: float = -1.1, lognormal_std: float = 2.0) -> Tensor:
    """consistency_model.py:141-146: discretised lognormal over the sigma intervals."""
    import numpy as np
    z = lambda s: torch.erf((torch.log(s) - lognormal_mean) / (lognormal_std * np.sqrt(2)))
    pdf = z(sigmas[1:]) - z(sigmas[:-1])
    return pdf / pdf.sum()


def ict_loss(net, sample: Tensor, sigmas: Tensor, timesteps: Tensor, epsilon: Tensor, cond=None,
             sigma_min: float = 0.002, sigma_data: float = 0.5) -> Tensor:
    """consistency_model.py:148-176 (eval-mode dropout; timesteps and epsilon injected): teacher at sigma_t without gradient,
    student at sigma_{t+1}, pseudo-Huber distance with c = 0.00054 sqrt(dim), weighted by 1 / (sigma_{t+1} - sigma_t)."""
    import numpy as np
    t_sig, s_sig = sigmas[timesteps], sigmas[timesteps + 1]
    with torch.no_grad():
        target = forward(net, sample + epsilon * _bcast(t_sig, sample.dim()), t_sig, None, cond, sigma_min, sigma_data)
    pred = forward(net, sample + epsilon * _bcast(s_sig, sample.dim()), s_sig, None, cond, sigma_min, sigma_data)
    c = 0.00054 * np.sqrt(np.prod(sample.shape[2:]))
    loss = torch.sqrt((pred - target) ** 2 + c**2) - c
    w = (1 / (sigmas[1:] - sigmas[:-1]))[timesteps]
    return (loss * _bcast(w, loss.dim())).mean()
