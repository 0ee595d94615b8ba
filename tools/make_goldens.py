#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *reference itself* in this container.

Run:  python tools/make_goldens.py            (needs /root/reference; CPU only)

The reference (highfem/tqdne, pure Python/PyTorch) is imported from
/root/reference -- never copied.  ``pytorch_lightning`` is not installed here, so a
tiny in-memory stand-in provides the few LightningModule attributes the hot-path
files touch (``save_hyperparameters``, ``device``, ``dtype``, ``log``) and
``isolate_rng``; it contributes no arithmetic.

What is captured (inputs + weights + outputs only -- data, not source):
  micro_unet.npz     UNetModel.forward on a micro config that exercises every code
                     path of the paper config (k=5 convs, k=3/s2 downsample, nearest-up
                     conv, attention with 2 heads, cond MLP, concat GroupNorm groups that
                     straddle the two concat sources), T=256 and the ragged T=248.
  micro_edm.npz      LightningEDM.forward at sigma in {0.002,0.5,80}, step() loss + grads,
                     sampling_sigmas(18), 18-step deterministic sample with intermediate
                     states, 6-step stochastic sample.
  micro_cm.npz       LithningConsistencyModel 1-step and 1-refinement sample.
  micro_ae.npz       LightningAutoencoder encode (injected eps) / decode.

Every all-zero parameter of a fresh model (zero_module: unet.py:102,357, blocks.py:134,249)
is re-drawn N(0, 0.02^2) and GroupNorm affines are jittered, otherwise a fresh model
outputs exactly zero and pins nothing.

All randomness is drawn from the CPU generator under a fixed seed and stored, so the
oracle / HIP side can have it injected.
"""

from __future__ import annotations

import contextlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def install_lightning_standin():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return next(self.parameters()).dtype

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    util = types.ModuleType("pytorch_lightning.utilities")
    seed = types.ModuleType("pytorch_lightning.utilities.seed")

    @contextlib.contextmanager
    def isolate_rng():
        st = torch.get_rng_state()
        try:
            yield
        finally:
            torch.set_rng_state(st)

    seed.isolate_rng = isolate_rng
    util.seed = seed
    pl.utilities = util
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = util
    sys.modules["pytorch_lightning.utilities.seed"] = seed


def perturb_(module: torch.nn.Module, seed: int):
    """Re-draw all-zero tensors and jitter GroupNorm affines (deterministic, CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            is_gn = p.ndim == 1 and (
                ".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name or name.startswith("out.0.")
                or ".out.0." in name
            )
            if is_gn and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif is_gn and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif torch.count_nonzero(p) == 0:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))


MICRO_UNET = dict(
    in_channels=3, out_channels=3, model_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1,
    attention_resolutions=(4,), num_heads=2, conv_kernel_size=5, dims=1, cond_features=5,
    dropout=0.0, flash_attention=False,
)
MICRO_AE = dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(), num_res_blocks=1,
                dims=1, conv_kernel_size=5, dropout=0.0)


def sd_np(module, prefix=""):
    return {"w:" + prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def main():
    sys.path.insert(0, REF)
    install_lightning_standin()
    torch.set_num_threads(8)
    from tqdne.unet import UNetModel
    from tqdne.edm import LightningEDM
    from tqdne.consistency_model import LithningConsistencyModel
    from tqdne.autoencoder import LightningAutoencoder

    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)

    # ---------------------------------------------------------------- UNet
    torch.manual_seed(0)
    net = UNetModel(**MICRO_UNET).eval()
    perturb_(net, 99)
    fx = dict(sd_np(net))
    taps_wanted = ["input_blocks.0", "input_blocks.1", "input_blocks.2", "input_blocks.5",
                   "middle_block", "output_blocks.1", "output_blocks.3", "output_blocks.5"]
    for T in (256, 248):
        x = torch.randn(2, 3, T, generator=g)
        t = torch.randn(2, generator=g) * 0.5
        c = torch.randn(2, 5, generator=g)
        caught = {}
        hooks = []
        for name in taps_wanted:
            mod = net.get_submodule(name)
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: caught.__setitem__(name, o.detach().clone())))
        with torch.no_grad():
            y = net(x, t, c)
        for h in hooks:
            h.remove()
        fx.update({f"T{T}:x": x.numpy(), f"T{T}:t": t.numpy(), f"T{T}:cond": c.numpy(), f"T{T}:y": y.numpy()})
        for k, v in caught.items():
            fx[f"T{T}:tap:{k}"] = v[:1].numpy()  # first sample only, keeps the fixture small
    fx["cfg"] = np.array(repr(MICRO_UNET))
    np.savez_compressed(os.path.join(OUT, "micro_unet.npz"), **fx)

    # ---------------------------------------------------------------- EDM
    torch.manual_seed(1)
    edm = LightningEDM(MICRO_UNET, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0},
                       num_sampling_steps=18).eval()
    edm.unet.load_state_dict(net.state_dict())  # weights live once, in micro_unet.npz
    fx = {}
    B, T = 2, 256
    sig = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g)
    fx.update(signal=sig.numpy(), cond=cond.numpy())
    for s in (0.002, 0.5, 80.0):
        sigma = torch.full((B,), s)
        xin = sig + s * torch.randn(B, 3, T, generator=g)
        with torch.no_grad():
            d = edm(xin, sigma, None, cond)
        fx[f"denoise:{s}:x"] = xin.numpy()
        fx[f"denoise:{s}:y"] = d.numpy()
    # loss + grads, randomness replicated by re-seeding the global generator
    torch.manual_seed(4242)
    eps = torch.randn(B)
    noise = torch.randn_like(sig)
    torch.manual_seed(4242)
    edm.zero_grad()
    loss = edm.step({"signal": sig, "cond": cond}, 0)
    loss.backward()
    fx.update({"step:eps": eps.numpy(), "step:noise": noise.numpy(), "step:loss": loss.detach().numpy()})
    for name in ["unet.input_blocks.0.0.weight", "unet.input_blocks.0.0.bias",
                 "unet.input_blocks.2.0.op.weight", "unet.middle_block.0.in_layers.2.weight",
                 "unet.middle_block.0.in_layers.0.weight", "unet.middle_block.0.in_layers.0.bias",
                 "unet.middle_block.1.qkv.weight", "unet.middle_block.1.proj_out.bias",
                 "unet.middle_block.0.emb_layers.1.weight", "unet.output_blocks.1.0.skip_connection.weight",
                 "unet.output_blocks.1.2.conv.weight", "unet.output_blocks.3.0.in_layers.0.weight",
                 "unet.out.2.weight", "unet.out.0.weight", "unet.time_mlp.0.weight", "unet.time_mlp.2.bias",
                 "unet.cond_mlp.0.weight", "unet.cond_mlp.2.weight"]:
        fx["step:grad:" + name] = edm.get_parameter(name).grad.numpy().copy()
    edm.zero_grad()
    fx["sigmas18"] = edm.edm.sampling_sigmas(18).numpy()
    # deterministic sampler, capturing the state after steps 1, 9, 18
    torch.manual_seed(777)
    start = torch.randn((B, 3, T), dtype=torch.float64)
    torch.manual_seed(777)
    trace = {}
    orig = edm.forward
    calls = {"n": 0}

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    edm.forward = counting
    with torch.no_grad():
        out = edm.sample((B, 3, T), cond=cond)
    edm.forward = orig
    fx.update({"sample:start": start.numpy(), "sample:out": out.numpy(), "sample:nfe": np.array(calls["n"])})
    for nsteps in (1, 9):
        # state after `nsteps` steps of the same 18-step schedule = run the loop truncated
        sigmas = edm.edm.sampling_sigmas(18)
        with torch.no_grad():
            st = _truncated(edm, start * sigmas[0], sigmas, cond, nsteps)
        fx[f"sample:state{nsteps}"] = st.numpy()
    # stochastic sampler, 6 steps
    edm.num_sampling_steps = 6
    edm.deterministic_sampling = False
    torch.manual_seed(888)
    s0 = torch.randn((B, 3, T), dtype=torch.float64)
    churn = [torch.randn((B, 3, T), dtype=torch.float64) for _ in range(6)]
    torch.manual_seed(888)
    with torch.no_grad():
        out = edm.sample((B, 3, T), cond=cond)
    fx.update({"stoch:start": s0.numpy(), "stoch:out": out.numpy(), "stoch:churn": np.stack([c.numpy() for c in churn])})
    fx["cfg"] = np.array(repr(MICRO_UNET))
    np.savez_compressed(os.path.join(OUT, "micro_edm.npz"), **fx)

    # ---------------------------------------------------------------- consistency
    cm = LithningConsistencyModel(net).eval()  # same weights as micro_unet.npz
    fx = {}
    cond = torch.randn(B, 5, generator=g)
    torch.manual_seed(555)
    e0 = torch.randn((B, 3, T))
    u0 = torch.rand((B, 3, T))
    torch.manual_seed(555)
    with torch.no_grad():
        y1 = cm.sample((B, 3, T), sigmas=[], cond=cond)
    torch.manual_seed(555)
    with torch.no_grad():
        y2 = cm.sample((B, 3, T), sigmas=[1.0], cond=cond)
    fx.update(cond=cond.numpy(), start=e0.numpy(), uniform=u0.numpy(), one_step=y1.numpy(), refined=y2.numpy())
    fx["cfg"] = np.array(repr(MICRO_UNET))
    np.savez_compressed(os.path.join(OUT, "micro_cm.npz"), **fx)

    # ---------------------------------------------------------------- autoencoder
    torch.manual_seed(3)
    enc_cfg = dict(MICRO_AE, in_channels=3, out_channels=8)
    dec_cfg = dict(MICRO_AE, in_channels=4, out_channels=3)
    ae = LightningAutoencoder(enc_cfg, dec_cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0}).eval()
    perturb_(ae, 13)
    fx = dict(sd_np(ae))
    x = 0.5 * torch.randn(2, 3, 256, generator=g)
    torch.manual_seed(321)
    eps = torch.randn(2, 4, 128)
    torch.manual_seed(321)
    with torch.no_grad():
        z, mean, log_std = ae._encode(x)
        xr = ae.decode(z)
    fx.update(x=x.numpy(), eps=eps.numpy(), z=z.numpy(), mean=mean.numpy(), log_std=log_std.numpy(), recon=xr.numpy())
    fx["enc_cfg"] = np.array(repr(enc_cfg))
    fx["dec_cfg"] = np.array(repr(dec_cfg))
    np.savez_compressed(os.path.join(OUT, "micro_ae.npz"), **fx)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


def _truncated(edm, eps, sigmas, cond, nsteps):
    """State after ``nsteps`` iterations of the reference's own loop, obtained by running the
    reference's sample_deterministically on a truncated schedule while keeping
    num_sampling_steps=18 (so the Heun-correction condition of edm.py:186 is unchanged)."""
    return edm.sample_deterministically(eps, sigmas[: nsteps + 1], None, cond)


if __name__ == "__main__":
    main()
