"""CPU restatement of the DDPM arithmetic behind tqdne/diffusion.py (test infrastructure only -- never imported by the product).

PARITY UNPINNED: the reference calls ``diffusers.DDPMScheduler`` (diffusion.py:3,77,98), which is neither vendored in
/root/reference nor installed here nor listed in the reference's lockfile, and the reference holds no vectors for this path.  The
functions restate the published algorithm -- Ho, Jain, Abbeel, "Denoising Diffusion Probabilistic Models" (NeurIPS 2020): eq. 4
(forward process), eq. 7 (posterior mean / variance), eq. 15 (x0 from epsilon), section 3.2 (sigma_t^2 = beta~_t) -- in float64,
independently of tqdne_amd/diffusion.py (no shared code), with the scheduler defaults diffusers documents.
  * module forward / loss .................. diffusion.py:55-65, 88-109
  * ancestral sampling loop ................ diffusion.py:67-79
"""

from __future__ import annotations

import numpy as np
import torch


def schedule(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
    betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32).astype(np.float64)
    return betas, np.cumprod(1.0 - betas)


def add_noise(x0, noise, t, abar):
    """eq. 4: x_t = sqrt(abar_t) x0 + sqrt(1 - abar_t) eps."""
    a = torch.as_tensor(np.sqrt(abar[t]), dtype=x0.dtype)
    c = torch.as_tensor(np.sqrt(1.0 - abar[t]), dtype=x0.dtype)
    sh = (-1,) + (1,) * (x0.dim() - 1)
    return a.reshape(sh) * x0 + c.reshape(sh) * noise


def loss(net, x0, noise, t, abar, prediction_type="epsilon", cond_signal=None, cond=None):
    """diffusion.py:88-109: MSE between the network output on x_t and the noise (or the clean sample)."""
    xt = add_noise(x0, noise, t, abar)
    x_in = xt if cond_signal is None else torch.cat((cond_signal, xt), dim=1)
    pred = net(x_in, torch.as_tensor(t, dtype=torch.float32), cond)
    target = noise if prediction_type == "epsilon" else x0
    return torch.mean((pred - target) ** 2)


def ancestral_step(x, model_out, t, prev_t, abar, z=None, prediction_type="epsilon", clip=1.0):
    """one reverse step: x0 from eq. 15, clipped; posterior mean of eq. 7; variance beta~_t = (1 - abar_prev) / (1 - abar_t) beta_t."""
    abar_t = abar[t]
    abar_prev = abar[prev_t] if prev_t >= 0 else 1.0
    alpha_t = abar_t / abar_prev
    beta_t = 1.0 - alpha_t
    x0 = (x - np.sqrt(1.0 - abar_t) * model_out) / np.sqrt(abar_t) if prediction_type == "epsilon" else model_out
    if clip:
        x0 = x0.clamp(-clip, clip)
    mean = (np.sqrt(abar_prev) * beta_t / (1.0 - abar_t)) * x0 + (np.sqrt(alpha_t) * (1.0 - abar_prev) / (1.0 - abar_t)) * x
    if t > 0:
        var = max((1.0 - abar_prev) / (1.0 - abar_t) * beta_t, 1e-20)
        mean = mean + np.sqrt(var) * z
    return mean


def sample(net, start, noises, timesteps, stride, abar, prediction_type="epsilon", cond_signal=None, cond=None, clip=1.0):
    """diffusion.py:67-79 with the draws injected; ``timesteps`` descending, ``stride`` = T // number of inference steps."""
    x = start
    for i, t in enumerate(timesteps):
        t = int(t)
        x_in = x if cond_signal is None else torch.cat((cond_signal, x), dim=1)
        pred = net(x_in.float(), torch.full((x.shape[0],), float(t)), cond).to(x.dtype)
        x = ancestral_step(x, pred, t, t - stride, abar, None if noises is None else noises[i], prediction_type, clip)
    return x
