"""CPU restatement of the EDM wrapper, loss and Heun samplers (test infrastructure only).

Reference lines restated (/root/reference/tqdne/edm.py):
  * EDM constants and scalar maps ........... edm.py:9-52
  * LightningEDM.forward (preconditioning) .. edm.py:105-113
  * LightningEDM.step (loss) ................ edm.py:115-134
  * sample / sample_deterministically ....... edm.py:146-196
  * sample_stochastically ................... edm.py:198-230

All randomness is *injected* (eps, noise, sampler start, churn noises) because the
HIP side cannot reproduce the CPU generator stream.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import unet as U

Tensor = torch.Tensor


@dataclass
class EDMParams:
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    rho: float = 7.0
    sigma_data: float = 0.5
    P_mean: float = -1.2
    P_std: float = 1.2
    S_churn: float = 40
    S_min: float = 0.05
    S_max: float = 50
    S_noise: float = 1.003


def sigma_of_eps(p: EDMParams, eps: Tensor) -> Tensor:  # edm.py:21-22
    return (eps * p.P_std + p.P_mean).exp()


def loss_weight(p: EDMParams, s: Tensor) -> Tensor:  # edm.py:24-25
    return (s**2 + p.sigma_data**2) / (s * p.sigma_data) ** 2


def c_skip(p: EDMParams, s: Tensor) -> Tensor:  # edm.py:27-28
    return p.sigma_data**2 / (s**2 + p.sigma_data**2)


def c_out(p: EDMParams, s: Tensor) -> Tensor:  # edm.py:30-31
    return s * p.sigma_data / (s**2 + p.sigma_data**2) ** 0.5


def c_in(p: EDMParams, s: Tensor) -> Tensor:  # edm.py:33-34
    return 1 / (s**2 + p.sigma_data**2) ** 0.5


def c_noise(p: EDMParams, s: Tensor) -> Tensor:  # edm.py:36-37
    return 0.25 * s.log()


def sampling_sigmas(p: EDMParams, num_steps: int) -> Tensor:  # edm.py:39-46
    ri = 1 / p.rho
    idx = torch.arange(num_steps, dtype=torch.float32)
    s = (p.sigma_max**ri + idx / (num_steps - 1) * (p.sigma_min**ri - p.sigma_max**ri)) ** p.rho
    return torch.cat([s, torch.zeros_like(s[:1])])


def sigma_hat(p: EDMParams, sigma: Tensor, num_steps: int) -> Tensor:  # edm.py:48-52
    gamma = min(p.S_churn / num_steps, 2**0.5 - 1) if p.S_min <= sigma <= p.S_max else 0
    return sigma + gamma * sigma


def _bcast(v: Tensor, ndim: int) -> Tensor:
    return v[(...,) + (None,) * (ndim - v.ndim)]


Net = Callable[[Tensor, Tensor, Optional[Tensor]], Tensor]


def make_net(sd, cfg, prefix: str = "unet.", dropout_masks=None) -> Net:
    def net(x, t, cond):
        return U.unet_forward(sd, cfg, x, t, cond, prefix=prefix, dropout_masks=dropout_masks)

    return net


def denoise(p: EDMParams, net: Net, sample: Tensor, sigma: Tensor, cond_sample=None, cond=None) -> Tensor:
    """edm.py:105-113."""
    nd = sample.dim()
    x_in = sample * _bcast(c_in(p, sigma), nd)
    if cond_sample is not None:
        x_in = torch.cat((x_in, cond_sample), dim=1)
    out = net(x_in, c_noise(p, sigma), cond)
    skip = _bcast(c_skip(p, sigma), nd) * sample
    return out * _bcast(c_out(p, sigma), nd) + skip


def loss_step(p: EDMParams, net: Net, signal: Tensor, eps: Tensor, unit_noise: Tensor, cond=None, cond_sample=None) -> Tensor:
    """edm.py:126-134 with eps ~ N(0,1)^B and unit_noise ~ N(0,1)^signal.shape injected."""
    sigma = sigma_of_eps(p, eps)
    noise = unit_noise * _bcast(sigma, signal.dim())
    pred = denoise(p, net, signal + noise, sigma, cond_sample, cond)
    loss = (pred - signal) ** 2
    return torch.mean(loss * _bcast(loss_weight(p, sigma), loss.dim()))


def sample_deterministic(
    p: EDMParams,
    net: Net,
    start_unit_noise: Tensor,
    num_steps: int,
    cond=None,
    cond_sample=None,
    net_dtype=torch.float32,
    trace: Optional[Dict[int, Tensor]] = None,
    stop_after: Optional[int] = None,
) -> Tensor:
    """edm.py:159-196.  ``start_unit_noise`` is the fp64 N(0,1) draw of edm.py:160
    (multiplied here by sigmas[0], an fp32 scalar, as the reference does).
    ``trace[i]`` receives the state after step i; ``stop_after`` ends the loop after that many steps (tests of the early
    steps at sizes where the full loop would take minutes on the CPU)."""
    dt = torch.float64
    sigmas = sampling_sigmas(p, num_steps)
    x_next = start_unit_noise.to(dt) * sigmas[0]
    n = len(start_unit_noise)
    for i, (s, s_next) in enumerate(zip(sigmas[:-1], sigmas[1:])):
        x = x_next
        d0 = denoise(p, net, x.to(net_dtype), s.to(net_dtype).repeat(n), cond_sample, cond).to(dt)
        d_cur = (x - d0) / s
        x_next = x + d_cur * (s_next - s)
        if i < num_steps - 1:
            d1 = denoise(p, net, x_next.to(net_dtype), s_next.to(net_dtype).repeat(n), cond_sample, cond).to(dt)
            d_prime = (x_next - d1) / s_next
            x_next = x + (s_next - s) * (0.5 * d_cur + 0.5 * d_prime)
        if trace is not None:
            trace[i + 1] = x_next.clone()
        if stop_after is not None and i + 1 >= stop_after:
            break
    return x_next


def sample_stochastic(
    p: EDMParams,
    net: Net,
    start_unit_noise: Tensor,
    churn_unit_noises: Sequence[Tensor],
    num_steps: int,
    cond=None,
    cond_sample=None,
    net_dtype=torch.float32,
) -> Tensor:
    """edm.py:198-230; ``churn_unit_noises[i]`` is the randn_like of edm.py:207 at step i."""
    dt = torch.float64
    sigmas = sampling_sigmas(p, num_steps)
    x_next = start_unit_noise.to(dt) * sigmas[0]
    n = len(start_unit_noise)
    for i, (s, s_next) in enumerate(zip(sigmas[:-1], sigmas[1:])):
        x = x_next
        s_hat = sigma_hat(p, s, num_steps)
        noise = churn_unit_noises[i].to(dt) * p.S_noise
        x_hat = x + noise * (s_hat**2 - s**2) ** 0.5
        d0 = denoise(p, net, x_hat.to(net_dtype), s_hat.to(net_dtype).repeat(n), cond_sample, cond).to(dt)
        d_cur = (x_hat - d0) / s_hat
        x_next = x_hat + d_cur * (s_next - s_hat)
        if i < num_steps - 1:
            d1 = denoise(p, net, x_next.to(net_dtype), s_next.to(net_dtype).repeat(n), cond_sample, cond).to(dt)
            d_prime = (x_next - d1) / s_next
            x_next = x_hat + (s_next - s_hat) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next
