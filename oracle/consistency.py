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
