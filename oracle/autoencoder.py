"""CPU restatement of the 1-D VAE encode/decode (test infrastructure only).

Reference lines restated:
  * Encoder / Decoder / un-conditioned ResBlock ... tqdne/blocks.py:233-436
  * LightningAutoencoder._encode / decode ........ tqdne/autoencoder.py:37-46
State-dict keys are the reference's (``encoder.input_layer``, ``encoder.down_blocks.{i}``,
``encoder.output_layer``, ``decoder.input_layer``, ``decoder.up_blocks.{i}``, ...).
"""

from __future__ import annotations

from typing import Optional

import torch

from . import unet as U

Tensor = torch.Tensor

AE_DEFAULTS = dict(
    attention_resolutions=(8, 16, 32), dropout=0, channel_mult=(1, 2, 4, 8),
    conv_kernel_size=3, conv_resample=True, dims=2, num_heads=1, flash_attention=True,
)


def _cfg(cfg):
    out = dict(AE_DEFAULTS)
    out.update(cfg)
    assert out["dims"] == 1 and out["conv_resample"]
    return out


def encoder_layout(cfg):
    """blocks.py:312-340 -> list of ("res"|"attn"|"down", index, cin, cout)."""
    cfg = _cfg(cfg)
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    ch = int(mult[0] * mc)
    ds, blocks = 1, []
    for level, m in enumerate(mult):
        for _ in range(cfg["num_res_blocks"]):
            blocks.append(("res", len(blocks), ch, int(m * mc)))
            ch = int(m * mc)
            if ds in cfg["attention_resolutions"]:
                blocks.append(("attn", len(blocks), ch, ch))
        if level != len(mult) - 1:
            blocks.append(("down", len(blocks), ch, ch))
            ds *= 2
    return blocks, ch


def decoder_layout(cfg):
    """blocks.py:400-430."""
    cfg = _cfg(cfg)
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    ch = int(mult[-1] * mc)
    ds, blocks = 2 ** (len(mult) - 1), []
    for level, m in reversed(list(enumerate(mult))):
        if level != len(mult) - 1:
            blocks.append(("up", len(blocks), ch, ch))
            ds //= 2
        for _ in range(cfg["num_res_blocks"]):
            blocks.append(("res", len(blocks), ch, int(m * mc)))
            ch = int(m * mc)
            if ds in cfg["attention_resolutions"]:
                blocks.append(("attn", len(blocks), ch, ch))
    return blocks, ch


def _run(sd, cfg, blocks, P, h):
    for kind, i, _, _ in blocks:
        p = f"{P}.{i}"
        if kind == "res":
            h = U.res_block(sd, p, h, None)
        elif kind == "attn":
            h = U.attention_block(sd, p, h, cfg["num_heads"])
        elif kind == "down":
            h = U.downsample(sd, p, h)
        else:
            h = U.upsample(sd, p, h)
    return h


def encoder_forward(sd, cfg, x: Tensor, prefix: str = "encoder.") -> Tensor:
    cfg = _cfg(cfg)
    blocks, _ = encoder_layout(cfg)
    h = U.conv_same(sd, prefix + "input_layer", x)
    h = _run(sd, cfg, blocks, prefix + "down_blocks", h)
    return U.conv_same(sd, prefix + "output_layer", h)


def decoder_forward(sd, cfg, z: Tensor, prefix: str = "decoder.") -> Tensor:
    cfg = _cfg(cfg)
    blocks, _ = decoder_layout(cfg)
    h = U.conv_same(sd, prefix + "input_layer", z)
    h = _run(sd, cfg, blocks, prefix + "up_blocks", h)
    return U.conv_same(sd, prefix + "output_layer", h)


def encode(sd, enc_cfg, x: Tensor, unit_noise: Tensor, prefix: str = "encoder."):
    """autoencoder.py:37-40: latent = mean + eps * exp(log_std), eps injected."""
    mean, log_std = torch.chunk(encoder_forward(sd, enc_cfg, x, prefix), 2, dim=1)
    return mean + unit_noise * torch.exp(log_std), mean, log_std


def decode(sd, dec_cfg, z: Tensor, prefix: str = "decoder.") -> Tensor:
    return decoder_forward(sd, dec_cfg, z, prefix)


def step_loss(sd, enc_cfg, dec_cfg, x: Tensor, unit_noise: Tensor, kl_weight: float = 1e-6):
    """autoencoder.py:59-66 (eval-mode dropout): loss = mean((x - D(z))^2) + kl_weight * mean(KL), with
    KL = 0.5 * sum_c(mean^2 + exp(2 log_std) - 2 log_std - 1) (autoencoder.py:54-57).  Returns (loss, recon_loss, kl)."""
    z, mean, log_std = encode(sd, enc_cfg, x, unit_noise)
    recon = decode(sd, dec_cfg, z)
    recon_loss = torch.mean((x - recon) ** 2)
    log_var = 2 * log_std
    kl = torch.mean(0.5 * torch.sum(mean ** 2 + torch.exp(log_var) - log_var - 1, dim=1))
    return recon_loss + kl_weight * kl, recon_loss, kl
