#!/usr/bin/env python3
"""Generate tests/golden/micro_ae_step.npz: the reference's LightningAutoencoder.step (autoencoder.py:59-84) -- loss, its two
terms and parameter gradients -- on the micro autoencoder of micro_ae.npz (same weights), with the randn_like draw injected.
Run:  python tools/make_ae_step_golden.py     (needs /root/reference; CPU only)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


def main():
    sys.path.insert(0, mg.REF)
    mg.install_lightning_standin()
    from tqdne.autoencoder import LightningAutoencoder

    z = np.load(os.path.join(mg.OUT, "micro_ae.npz"))
    enc_cfg = eval(str(z["enc_cfg"]), {"__builtins__": {}}, {"dict": dict})
    dec_cfg = eval(str(z["dec_cfg"]), {"__builtins__": {}}, {"dict": dict})
    ae = LightningAutoencoder(enc_cfg, dec_cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0}, kl_weight=1e-2).eval()
    ae.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    g = torch.Generator().manual_seed(77)
    x = 0.5 * torch.randn(2, 3, 256, generator=g)
    cx = 0.3 * torch.randn(2, 3, 256, generator=g)
    draws = [torch.randn(2, 4, 128, generator=g), torch.randn(2, 4, 128, generator=g)]
    it = iter(draws)
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: next(it)  # the two draws of _encode (signal, cond_signal), autoencoder.py:39
    try:
        loss = ae.step({"signal": x, "cond_signal": cx})
    finally:
        torch.randn_like = orig
    loss.backward()
    it = iter(draws[:1])
    torch.randn_like = lambda t, **k: next(it)
    try:
        ae2 = LightningAutoencoder(enc_cfg, dec_cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0}, kl_weight=1e-2).eval()
        ae2.load_state_dict(ae.state_dict())
        loss1 = ae2.step({"signal": x})
    finally:
        torch.randn_like = orig
    loss1.backward()
    fx = dict(x=x.numpy(), cond_x=cx.numpy(), eps0=draws[0].numpy(), eps1=draws[1].numpy(), loss=loss.detach().numpy(),
              loss_signal_only=loss1.detach().numpy(), kl_weight=np.array(1e-2))
    for n, p in ae.named_parameters():
        fx["g:" + n] = p.grad.numpy()
    for n, p in ae2.named_parameters():
        if n in ("encoder.input_layer.weight", "decoder.input_layer.weight", "decoder.output_layer.weight", "encoder.output_layer.bias"):
            fx["g1:" + n] = p.grad.numpy()
    np.savez_compressed(os.path.join(mg.OUT, "micro_ae_step.npz"), **fx)
    print("loss", float(loss), float(loss1), os.path.getsize(os.path.join(mg.OUT, "micro_ae_step.npz")))


if __name__ == "__main__":
    main()
