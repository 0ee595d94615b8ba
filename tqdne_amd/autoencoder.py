"""1-D VAE (latent EDM, BASELINE config 3): drop-in for the encode / decode part of ``tqdne.autoencoder`` and the
``Encoder`` / ``Decoder`` of ``tqdne.blocks`` (reference tqdne/autoencoder.py:9-49, tqdne/blocks.py:233-436), on the same
fused HIP kernels as the UNet (un-conditioned ResBlock = the fused convs without the embedding add).

Same constructor keywords and ``state_dict`` schema (``encoder.input_layer``, ``encoder.down_blocks.{i}.*``,
``encoder.output_layer``, ``decoder.input_layer``, ``decoder.up_blocks.{i}.*``, ``decoder.output_layer``).

Training (``step``: MSE + KL, autoencoder.py:59-84) runs the HIP forward and backward of both networks; the re-parameterisation
and the two scalar losses are a few elementwise torch ops on the (B, 2 x latent, T / ds) encoder output (1/100 of a conv).
"""

from __future__ import annotations

import torch as th
from torch import nn

from ._cache import plan_cache
from . import engine, rng
from .lightning_compat import LightningModule
from .unet import AttentionParams, DownsampleParams, ResBlockParams, UpsampleParams, _conv


def _check(dims, conv_resample):
    if dims not in (1, 2) or not conv_resample:
        raise NotImplementedError("tqdne_amd autoencoder: conv-resample, dims=1 (HIP path) or dims=2 (stock-PyTorch family)")
    if dims == 2:
        from . import family2d
        family2d.announce()


class Encoder(nn.Module):
    """blocks.py:263-348."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions=(8, 16, 32),
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_kernel_size=3, conv_resample=True, dims=2, num_heads=1,
                 flash_attention=True):
        super().__init__()
        _check(dims, conv_resample)
        self.in_channels, self.out_channels, self.num_heads = in_channels, out_channels, num_heads
        self.dropout, self.dims = dropout, dims
        k = conv_kernel_size
        ch = int(channel_mult[0] * model_channels)
        self.input_layer = _conv(dims)(in_channels, ch, k, padding="same")
        ds, blocks = 1, []
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                blocks.append(ResBlockParams(ch, None, dropout, int(mult * model_channels), k, dims))
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    blocks.append(AttentionParams(ch, num_heads, dims))
            if level != len(channel_mult) - 1:
                blocks.append(DownsampleParams(ch, ch, dims=dims))  # kernel 3 (blocks.py:337 passes none)
                ds *= 2
        self.down_blocks = nn.Sequential(*blocks)
        self.output_layer = _conv(dims)(ch, out_channels, k, padding="same")
        self.time_scale = ds  # T_out = T_in / ds
        self._engine_cache = plan_cache()

    blocks_attr = "down_blocks"

    def forward(self, x):
        if self.dims == 2:   # stock PyTorch operators (family2d.py): the generate_waveforms.py family only
            from . import family2d
            return family2d.coder_forward(self, x)
        engine.require_device(x)
        eng = _seq_engine(self, x)
        y = eng.forward(x).clone()
        if eng.check_range():   # activations near the fp16 range: the plan is on bf16x3 now, repeat
            y = eng.forward(x).clone()
        return y

    def _apply(self, fn, *a, **k):
        self._engine_cache = plan_cache()
        self.__dict__.pop("_packed_stores", None)
        return super()._apply(fn, *a, **k)


class Decoder(nn.Module):
    """blocks.py:351-436."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions=(8, 16, 32),
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_kernel_size=3, conv_resample=True, dims=2, num_heads=1,
                 flash_attention=True):
        super().__init__()
        _check(dims, conv_resample)
        self.in_channels, self.out_channels, self.num_heads = in_channels, out_channels, num_heads
        self.dropout, self.dims = dropout, dims
        k = conv_kernel_size
        ch = int(channel_mult[-1] * model_channels)
        self.input_layer = _conv(dims)(in_channels, ch, k, padding="same")
        ds, blocks = 2 ** (len(channel_mult) - 1), []
        for level, mult in reversed(list(enumerate(channel_mult))):
            if level != len(channel_mult) - 1:
                blocks.append(UpsampleParams(ch, ch, dims=dims))  # kernel 3 (blocks.py:408 passes none)
                ds //= 2
            for _ in range(num_res_blocks):
                blocks.append(ResBlockParams(ch, None, dropout, int(mult * model_channels), k, dims))
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    blocks.append(AttentionParams(ch, num_heads, dims))
        self.up_blocks = nn.Sequential(*blocks)
        self.output_layer = _conv(dims)(ch, out_channels, k, padding="same")
        self._engine_cache = plan_cache()

    blocks_attr = "up_blocks"

    def forward(self, x):
        if self.dims == 2:   # stock PyTorch operators (family2d.py): the generate_waveforms.py family only
            from . import family2d
            return family2d.coder_forward(self, x)
        engine.require_device(x)
        eng = _seq_engine(self, x)
        y = eng.forward(x).clone()
        if eng.check_range():   # activations near the fp16 range: the plan is on bf16x3 now, repeat
            y = eng.forward(x).clone()
        return y

    def _apply(self, fn, *a, **k):
        self._engine_cache = plan_cache()
        self.__dict__.pop("_packed_stores", None)
        return super()._apply(fn, *a, **k)


def _seq_engine(mod, x):
    key = (x.shape[0], x.shape[2], str(x.device))
    eng = mod._engine_cache.get(key)
    if eng is None:
        eng = engine.SeqEngine(mod, x.shape[0], x.shape[2], x.device)
        mod._engine_cache[key] = eng
    return eng




class _AELossFn(th.autograd.Function):
    """loss = mean((x - D(z))^2) + kl_weight * KL(q(z|x) || N(0, I)),  z = mean + eps * exp(log_std)   (autoencoder.py:59-66).
    The HIP backward runs eagerly inside ``forward`` (static engine buffers would not survive a second pass, e.g. the
    ``cond_signal`` term); ``backward`` only scales the stored gradients."""

    @staticmethod
    def forward(ctx, module, x, stage, prefix, *params):
        need = any(ctx.needs_input_grad[4:])  # (grad mode is off inside forward; this reflects the caller's)
        rl, kl, grads = module._loss_and_grads(x, want_grads=need)
        loss = rl + module.kl_weight * kl
        module.log(f"{stage}/{prefix}reconstruction_loss", rl.item(), sync_dist=True)
        module.log(f"{stage}/{prefix}kl_divergence", kl.item(), sync_dist=True)
        module.log(f"{stage}/{prefix}loss", loss.item(), sync_dist=True)
        ctx.grads = grads
        return loss.clone()

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grads is None:
            raise RuntimeError("the loss was computed without gradients")
        return (None, None, None, None) + tuple(None if g is None else g * gloss for g in ctx.grads)


class LightningAutoencoder(LightningModule):
    def __init__(self, encoder_config: dict, decoder_config: dict, optimizer_params: dict, kl_weight: float = 1e-6):
        super().__init__()
        self.encoder = Encoder(**encoder_config)
        self.decoder = Decoder(**decoder_config)
        self.optimizer_params = optimizer_params
        self.kl_weight = kl_weight
        self.config = encoder_config
        self.save_hyperparameters()

    def _encode(self, x, unit_noise=None):
        """autoencoder.py:37-40.  ``unit_noise`` (optional) injects the N(0,1) draw of ``randn_like(mean)``."""
        from . import _lib
        enc = self.encoder(x)
        mean, log_std = th.chunk(enc, 2, dim=1)
        if self.encoder.dims == 2:   # stock-PyTorch family (family2d.py)
            eps = th.randn_like(mean) if unit_noise is None else unit_noise
            return mean + eps * th.exp(log_std), mean, log_std
        eps = (th.randn_like(mean) if unit_noise is None else unit_noise).contiguous().float()
        latent = th.empty_like(eps)
        B, L, Tl = eps.shape
        _lib.check(_lib.load().tq_vae_reparam_fwd(enc.data_ptr(), eps.data_ptr(), latent.data_ptr(), None, B, L, Tl,
                                                  th.cuda.current_stream(x.device).cuda_stream), "vae reparam")
        return latent, mean, log_std

    def encode(self, x):
        return self._encode(x)[0]

    def decode(self, x):
        return self.decoder(x.contiguous())   # (dims=2: Decoder.forward dispatches to family2d)

    def forward(self, x):
        return self.decode(self._encode(x)[0])

    def evaluate(self, batch):
        return self(batch["signal"])

    def kl_divergence(self, mean, log_std):
        log_var = 2 * log_std
        return 0.5 * th.sum(mean**2 + th.exp(log_var) - log_var - 1, dim=1)

    # ------------------------------------------------------------------ training (autoencoder.py:59-84)
    def _loss_and_grads(self, x, unit_noise=None, want_grads=True):
        """recon_loss, kl, and (if wanted) the gradients of recon_loss + kl_weight * kl wrt every parameter, in
        ``self.parameters()`` order.  Forward and backward run back to back: the engines' buffers are static."""
        engine.require_device(x)
        x = x.contiguous()
        train = self.training
        seed = rng.next_dropout_seed()
        from . import _lib
        from ._lib import check
        lib = _lib.load()
        _p = lambda t: None if t is None else t.data_ptr()
        with th.no_grad():
            stream = th.cuda.current_stream(x.device).cuda_stream
            e_eng = _seq_engine(self.encoder, x)
            enc = e_eng.forward(x, train=train, dropout_seed=seed)            # (B, 2L, T') = [mean | log_std], static buffer
            B, L2, Tl = enc.shape
            L = L2 // 2
            key = ("ae", tuple(enc.shape), str(x.device))
            bufs = getattr(self, "_ae_bufs", {}).get(key)
            if bufs is None:
                f = lambda *sh: th.empty(*sh, device=x.device)
                bufs = dict(z=f(B, L, Tl), kl=f(1), rl=f(1), drecon=th.empty_like(x), denc=f(B, L2, Tl))
                self.__dict__.setdefault("_ae_bufs", {})[key] = bufs
            eps = (th.randn_like(enc[:, :L]) if unit_noise is None else unit_noise).contiguous()  # (the draw of autoencoder.py:39)
            # z = mean + eps * exp(log_std) and the KL term in one launch (autoencoder.py:37-43, 64-66)
            check(lib.tq_vae_reparam_fwd(_p(enc), _p(eps), _p(bufs["z"]), _p(bufs["kl"]), B, L, Tl, stream), "vae reparam")
            d_eng = _seq_engine(self.decoder, bufs["z"])
            recon = d_eng.forward(bufs["z"], train=train, dropout_seed=seed ^ 0x9E3779B97F4A7C15)
            check(lib.tq_mse_loss(_p(recon), _p(x), _p(bufs["rl"]), _p(bufs["drecon"]) if want_grads else None, x.numel(), stream),
                  "mse loss")
            recon_loss, kl = bufs["rl"][0].clone(), bufs["kl"][0].clone()
            if not want_grads:
                return recon_loss, kl, None
            g_dec, dz = d_eng.backward(bufs["drecon"], want_dx=True, clone=True)
            # d loss / d [mean | log_std] from d z and the KL term (kl is a mean over (batch, time) of a sum over channels)
            check(lib.tq_vae_reparam_bwd(_p(enc), _p(eps), _p(dz.contiguous()), _p(bufs["denc"]), float(self.kl_weight), B, L, Tl, stream),
                  "vae reparam bwd")
            g_enc, _ = e_eng.backward(bufs["denc"], want_dx=False, clone=True)
        return recon_loss, kl, list(g_enc) + list(g_dec)

    def _step_2d(self, x, stage, prefix):
        """autoencoder.py:59-69 on torch operators and torch.autograd (dims=2 family)."""
        latent, mean, log_std = self._encode(x)
        recon_loss = th.mean((x - self.decode(latent)) ** 2)
        kl = th.mean(self.kl_divergence(mean, log_std))
        loss = recon_loss + self.kl_weight * kl
        self.log(f"{stage}/{prefix}reconstruction_loss", recon_loss.detach().item(), sync_dist=True)
        self.log(f"{stage}/{prefix}kl_divergence", kl.detach().item(), sync_dist=True)
        self.log(f"{stage}/{prefix}loss", loss.detach().item(), sync_dist=True)
        return loss

    def step(self, batch, stage="training"):
        x = batch["signal"]
        if self.encoder.dims == 2:
            loss = self._step_2d(x, stage, "")
            return loss if "cond_signal" not in batch else loss + self._step_2d(batch["cond_signal"], stage, "cond_")
        loss = _AELossFn.apply(self, x, stage, "", *self.parameters())
        if "cond_signal" not in batch:
            return loss
        return loss + _AELossFn.apply(self, batch["cond_signal"], stage, "cond_", *self.parameters())

    def step_and_backward(self, batch):
        """``step`` + backward without the autograd round trip: gradients are left in ``p.grad`` (DataParallelTrainer)."""
        if self.encoder.dims == 2:
            raise NotImplementedError("DataParallelTrainer drives the 1-D HIP path; train dims=2 models with step() + torch.autograd")
        xs = [batch["signal"]] + ([batch["cond_signal"]] if "cond_signal" in batch else [])
        total, grads = None, None
        for x in xs:
            rl, kl, g = self._loss_and_grads(x)
            loss = rl + self.kl_weight * kl
            total = loss if total is None else total + loss
            grads = g if grads is None else [a if b is None else (b if a is None else a + b) for a, b in zip(grads, g)]
        flats = []
        for p, g in zip(self.parameters(), grads):
            if g is not None:
                p.grad = g
                flats.append(g)
        return total, flats

    def training_step(self, batch, batch_idx):
        return self.step(batch, stage="training")

    def validation_step(self, batch, batch_idx):
        return self.step(batch, stage="validation")

    def configure_optimizers(self):
        optimizer = th.optim.AdamW(self.parameters(), lr=self.optimizer_params["learning_rate"], weight_decay=1e-4)
        lr_scheduler = th.optim.lr_scheduler.CosineAnnealingLR(
            optimizer, T_max=self.optimizer_params["max_steps"], eta_min=self.optimizer_params["eta_min"])
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": lr_scheduler, "interval": "step"}}
