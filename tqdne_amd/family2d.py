"""The 2-D (spectrogram) model family on stock PyTorch operators -- SURVEY.md section 8 row N4's second half.

The reference's published CLI (``tqdne/generate_waveforms.py:118-193``) is hard-wired to ``dims=2`` models
(``architectures.py:40-79``): a latent EDM over 128 x 128 log-spectrograms.  That family is NOT the hot path this
package accelerates and has no HIP kernels; SURVEY.md asks for it to keep working "through the
stock-PyTorch fallback" so that a user of the reference finds the CLI's models loadable and runnable.  This module is that
fallback and nothing else:

* it is reached only by models constructed with ``dims=2`` -- a ``dims=1`` model never comes here and still fails loudly
  without the HIP library or on CPU tensors (``engine.require_device``);
* the parameter containers of ``unet.py`` / ``autoencoder.py`` are ordinary ``torch.nn`` modules, so the arithmetic below is
  their own ``forward`` (``nn.Conv2d``, ``nn.GroupNorm``, ``nn.Linear``) wired the way the reference wires them; it runs on
  whatever device the tensors live on and is differentiable by ``torch.autograd``;
* a one-time warning says so when the first 2-D model is built.

Wiring followed (behaviour, not code): ``tqdne/unet.py:127-143,360-398`` (ResBlock, UNet), ``blocks.py:15-26`` (Fourier
features), ``blocks.py:56-66,100-108`` (resampling), ``blocks.py:136-190`` (attention over the flattened positions),
``blocks.py:345-360,433-450`` (Encoder / Decoder), ``edm.py:105-134,171-230`` (preconditioning, loss, Heun samplers).
"""

from __future__ import annotations

import math
import warnings
from typing import Callable, Optional

import torch as th
import torch.nn.functional as F

_warned = False


def announce():
    global _warned
    if not _warned:
        _warned = True
        warnings.warn("tqdne_amd: dims=2 models run on stock PyTorch operators (SURVEY.md 8 N4 fallback for the reference's "
                      "generate_waveforms.py family); only dims=1 models use the HIP kernels", stacklevel=3)


def _bcast(v: th.Tensor, like: th.Tensor) -> th.Tensor:
    """(N,) -> (N, 1, 1, ...) with ``like``'s rank (reference nn.py:78-83)."""
    return v.reshape(v.shape + (1,) * (like.dim() - v.dim()))


# ------------------------------------------------------------------------------------------------ layers
def fourier_features(W: th.Tensor, t: th.Tensor) -> th.Tensor:
    ang = ((t[:, None] * W[None, :]) * 2) * math.pi          # blocks.py:23 (this grouping, fp32)
    return th.cat([ang.sin(), ang.cos()], dim=-1)


def res_block(rb, x: th.Tensor, emb: Optional[th.Tensor]) -> th.Tensor:
    h = rb.in_layers(x)
    if emb is not None:
        h = h + _bcast(rb.emb_layers(emb).to(h.dtype), h)   # unet.py:131-139 (additive conditioning only)
    return rb.skip_connection(x) + rb.out_layers(h)


def attention_block(ab, x: th.Tensor) -> th.Tensor:
    n, c = x.shape[:2]
    qkv = ab.qkv(ab.norm(x)).reshape(n, 3 * c, -1)
    heads = ab.num_heads
    d = c // heads
    q, k, v = (t.reshape(n * heads, d, -1) for t in qkv.chunk(3, dim=1))
    s = 1.0 / math.sqrt(math.sqrt(d))                          # both operands scaled by d^-1/4 (blocks.py:171-178)
    w = th.softmax(th.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1).to(q.dtype)
    a = th.einsum("bts,bcs->bct", w, v).reshape(x.shape)
    return x + ab.proj_out(a)


def apply_layer(layer, x: th.Tensor, emb: Optional[th.Tensor]) -> th.Tensor:
    kind = getattr(layer, "kind", None)
    if kind == "res":
        return res_block(layer, x, emb)
    if kind == "attn":
        return attention_block(layer, x)
    if kind == "down":
        return layer.op(x)
    if kind == "up":
        return layer.conv(F.interpolate(x, scale_factor=2, mode="nearest"))
    return layer(x)   # the stem convolution


def unet_forward(m, x: th.Tensor, timesteps: th.Tensor, cond: Optional[th.Tensor]) -> th.Tensor:
    emb = m.time_mlp(fourier_features(m.time_embed.W, timesteps))
    if m.cond_features is not None:
        emb = emb + m.cond_mlp(cond)
    h = x
    saved = []
    for block in m.input_blocks:
        for layer in block:
            h = apply_layer(layer, h, emb)
        saved.append(h)
    for layer in m.middle_block:
        h = apply_layer(layer, h, emb)
    for block in m.output_blocks:
        h = th.cat([h, saved.pop()], dim=1)
        for layer in block:
            h = apply_layer(layer, h, emb)
    return m.out(h)


def coder_forward(m, x: th.Tensor) -> th.Tensor:
    """Encoder / Decoder of the autoencoder: stem, the block sequence (no embedding), output conv."""
    h = m.input_layer(x)
    for layer in getattr(m, m.blocks_attr):
        h = apply_layer(layer, h, None)
    return m.output_layer(h)


# ------------------------------------------------------------------------------------------------ EDM
def denoise(module, sample: th.Tensor, sigma: th.Tensor, cond_sample=None, cond=None) -> th.Tensor:
    e = module.edm
    x_in = sample * _bcast(e.in_scaling(sigma), sample)
    if cond_sample is not None:
        x_in = th.cat((x_in, cond_sample), dim=1)
    out = module.unet(x_in, e.noise_conditioning(sigma), cond=cond)
    return out * _bcast(e.out_scaling(sigma), sample) + _bcast(e.skip_scaling(sigma), sample) * sample


def edm_loss(module, sample: th.Tensor, eps: th.Tensor, unit_noise: th.Tensor, cond=None, cond_sample=None) -> th.Tensor:
    sigma = module.edm.sigma(eps)
    pred = denoise(module, sample + unit_noise * _bcast(sigma, sample), sigma, cond_sample, cond)
    return th.mean((pred - sample) ** 2 * _bcast(module.edm.loss_weight(sigma), sample))


def heun_sample(module, start: th.Tensor, sigmas: th.Tensor, cond_sample=None, cond=None,
                churn: Optional[Callable[[th.Tensor], th.Tensor]] = None) -> th.Tensor:
    """Heun's second-order integration of the probability-flow ODE in the start state's dtype (fp64), network calls in the
    parameters' dtype.  ``churn(state)`` -> unit noise switches the per-step noise injection on (edm.py:198-230);
    without it this is the deterministic sampler (edm.py:171-196)."""
    net_dtype = next(module.unet.parameters()).dtype
    steps = module.num_sampling_steps
    acc = start.dtype

    def D(x, s):
        return denoise(module, x.to(net_dtype), s.to(net_dtype).repeat(len(x)), cond_sample, cond).to(acc)

    x = start
    for i in range(len(sigmas) - 1):
        s_from, s_to = sigmas[i], sigmas[i + 1]
        if churn is not None:
            s_hat = module.edm.sigma_hat(s_from, steps)
            x = x + churn(x) * module.edm.S_noise * (s_hat**2 - s_from**2) ** 0.5
            s_from = s_hat
        slope = (x - D(x, s_from)) / s_from
        x_euler = x + slope * (s_to - s_from)
        if i < steps - 1:
            slope_to = (x_euler - D(x_euler, s_to)) / s_to
            x = x + (s_to - s_from) * (0.5 * slope + 0.5 * slope_to)
        else:
            x = x_euler
    return x
