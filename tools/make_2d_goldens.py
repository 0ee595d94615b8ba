#!/usr/bin/env python3
"""Generate tests/golden/micro_2d.npz by running the reference's dims=2 models in this container (CPU).

Run:  python tools/make_2d_goldens.py        (needs /root/reference; imports it, copies nothing)

SURVEY.md section 8 row N4: the reference's generate_waveforms.py builds a 2-D latent EDM (architectures.py:40-79).  The fixture
pins tqdne_amd's stock-PyTorch 2-D family (tqdne_amd/family2d.py) to the reference on a micro configuration that has every
layer kind of the real one (3x3 convs, stride-2 down-sampling, nearest up-sampling, attention over the flattened positions,
conditioning MLP, concat skips):
  UNetModel.forward, LightningEDM.forward at three noise levels, step() loss + a few gradients (injected noise),
  4-step deterministic and 3-step stochastic Heun samples, LightningAutoencoder encode / decode, and the latent pipeline
  (encode-shape inference, latent sampling, decode).
All-zero parameters (zero_module) are re-drawn and GroupNorm affines jittered, as in tools/make_goldens.py.
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, install_lightning_standin, perturb_, sd_np  # noqa: E402

UNET_2D = dict(in_channels=4, out_channels=4, model_channels=32, channel_mult=(1, 2), num_res_blocks=1,
               attention_resolutions=(2,), num_heads=2, conv_kernel_size=3, dims=2, cond_features=5, dropout=0.0,
               flash_attention=False)
AE_2D = dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(), num_res_blocks=1, dims=2, conv_kernel_size=3,
             dropout=0.0)


def main():
    sys.path.insert(0, REF)
    install_lightning_standin()
    torch.set_num_threads(8)
    from tqdne.autoencoder import LightningAutoencoder
    from tqdne.edm import LightningEDM
    from tqdne.unet import UNetModel

    g = torch.Generator().manual_seed(2024)
    opt = {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}
    fx = {}

    torch.manual_seed(0)
    net = UNetModel(**UNET_2D).eval()
    perturb_(net, 7)
    fx.update(sd_np(net, "unet."))
    B, H, W = 2, 16, 24   # (not square: catches a swapped axis)
    x = torch.randn(B, 4, H, W, generator=g)
    t = 0.5 * torch.randn(B, generator=g)
    cond = torch.randn(B, 5, generator=g)
    with torch.no_grad():
        y = net(x, t, cond)
    fx.update({"unet:x": x.numpy(), "unet:t": t.numpy(), "unet:cond": cond.numpy(), "unet:y": y.numpy()})

    edm = LightningEDM(UNET_2D, opt, num_sampling_steps=4).eval()
    edm.unet.load_state_dict(net.state_dict())
    sig = 0.5 * torch.randn(B, 4, H, W, generator=g)
    fx["edm:signal"] = sig.numpy()
    for s in (0.002, 0.5, 80.0):
        xin = sig + s * torch.randn(B, 4, H, W, generator=g)
        with torch.no_grad():
            d = edm(xin, torch.full((B,), s), None, cond)
        fx[f"edm:denoise:{s}:x"] = xin.numpy()
        fx[f"edm:denoise:{s}:y"] = d.numpy()
    torch.manual_seed(11)
    eps = torch.randn(B)
    noise = torch.randn_like(sig)
    torch.manual_seed(11)
    edm.zero_grad()
    loss = edm.step({"signal": sig, "cond": cond}, 0)
    loss.backward()
    fx.update({"edm:step:eps": eps.numpy(), "edm:step:noise": noise.numpy(), "edm:step:loss": loss.detach().numpy()})
    for name in ("unet.input_blocks.0.0.weight", "unet.input_blocks.2.0.op.weight", "unet.middle_block.1.qkv.weight",
                 "unet.output_blocks.1.0.skip_connection.weight", "unet.out.2.weight", "unet.cond_mlp.0.weight"):
        fx["edm:step:grad:" + name] = edm.get_parameter(name).grad.numpy().copy()
    edm.zero_grad()
    torch.manual_seed(21)
    start = torch.randn((B, 4, H, W), dtype=torch.float64)
    torch.manual_seed(21)
    with torch.no_grad():
        out = edm.sample((B, 4, H, W), cond=cond)
    fx.update({"edm:sample:start": start.numpy(), "edm:sample:out": out.numpy()})
    edm.num_sampling_steps, edm.deterministic_sampling = 3, False
    torch.manual_seed(22)
    s0 = torch.randn((B, 4, H, W), dtype=torch.float64)
    churn = [torch.randn((B, 4, H, W), dtype=torch.float64) for _ in range(3)]
    torch.manual_seed(22)
    with torch.no_grad():
        out = edm.sample((B, 4, H, W), cond=cond)
    fx.update({"edm:stoch:start": s0.numpy(), "edm:stoch:churn": np.stack([c.numpy() for c in churn]), "edm:stoch:out": out.numpy()})

    torch.manual_seed(3)
    enc_cfg = dict(AE_2D, in_channels=3, out_channels=8)
    dec_cfg = dict(AE_2D, in_channels=4, out_channels=3)
    ae = LightningAutoencoder(enc_cfg, dec_cfg, opt).eval()
    perturb_(ae, 13)
    fx.update(sd_np(ae, "ae."))
    xa = 0.5 * torch.randn(B, 3, 32, 48, generator=g)
    torch.manual_seed(31)
    e_draw = torch.randn(B, 4, 16, 24)
    torch.manual_seed(31)
    with torch.no_grad():
        z, mean, log_std = ae._encode(xa)
        recon = ae.decode(z)
    fx.update({"ae:x": xa.numpy(), "ae:eps": e_draw.numpy(), "ae:z": z.numpy(), "ae:mean": mean.numpy(),
               "ae:log_std": log_std.numpy(), "ae:recon": recon.numpy()})
    torch.manual_seed(32)
    e2 = torch.randn(B, 4, 16, 24)
    torch.manual_seed(32)
    ae.train()
    l_ae = ae.step({"signal": xa})
    ae.eval()
    fx.update({"ae:step:eps": e2.numpy(), "ae:step:loss": l_ae.detach().numpy()})

    # latent pipeline: sample() encodes a zeros tensor for the shape (one randn draw in _encode), then draws the start state
    # (the 4-channel UNet above doubles as the latent denoiser: weights stored once)
    ledm = LightningEDM(UNET_2D, opt, num_sampling_steps=3, autoencoder=ae).eval()
    ledm.unet.load_state_dict(net.state_dict())
    torch.manual_seed(41)
    _ = torch.randn(B, 4, 16, 24)                       # the draw inside the shape-inference encode (edm.py:154-157)
    lstart = torch.randn((B, 4, 16, 24), dtype=torch.float64)
    torch.manual_seed(41)
    with torch.no_grad():
        lout = ledm.sample((B, 3, 32, 48), cond=cond)
    fx.update({"latent:start": lstart.numpy(), "latent:out": lout.numpy()})

    fx["unet_cfg"] = np.array(repr(UNET_2D))
    fx["enc_cfg"] = np.array(repr(enc_cfg))
    fx["dec_cfg"] = np.array(repr(dec_cfg))
    path = os.path.join(OUT, "micro_2d.npz")
    np.savez_compressed(path, **fx)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
