#!/usr/bin/env python3
"""Generate tests/golden/micro_cm_step.npz: the reference's iCT training step (consistency_model.py:115-176) on the micro UNet
of micro_unet.npz -- loss and parameter gradients -- with the two random draws (torch.multinomial, randn_like) injected and the
Lightning attributes the step reads (trainer.max_steps, global_step) set by hand.
Run:  python tools/make_cm_step_golden.py     (needs /root/reference; CPU only)"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


def main():
    sys.path.insert(0, mg.REF)
    mg.install_lightning_standin()
    from tqdne.consistency_model import LithningConsistencyModel
    from tqdne.unet import UNetModel

    z = np.load(os.path.join(mg.OUT, "micro_unet.npz"))
    net = UNetModel(**mg.MICRO_UNET).eval()
    net.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    cm = LithningConsistencyModel(net).eval()
    cm.trainer = types.SimpleNamespace(max_steps=4000)
    cm.global_step = 1700
    g = torch.Generator().manual_seed(909)
    B, T = 2, 256
    sample = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g)
    eps = torch.randn(B, 3, T, generator=g)
    captured = {}
    o_mult, o_randn = torch.multinomial, torch.randn_like

    def mult(pdf, n, replacement=True):
        captured["pdf"] = pdf.clone()
        t = torch.tensor([3, len(pdf) - 1][:n]) if n <= 2 else o_mult(pdf, n, replacement)
        captured["timesteps"] = t.clone()
        return t

    torch.multinomial, torch.randn_like = mult, (lambda t, **k: eps)
    try:
        loss = cm.step({"signal": sample, "cond": cond})
    finally:
        torch.multinomial, torch.randn_like = o_mult, o_randn
    loss.backward()
    fx = dict(sample=sample.numpy(), cond=cond.numpy(), eps=eps.numpy(), timesteps=captured["timesteps"].numpy(),
              pdf=captured["pdf"].numpy(), loss=loss.detach().numpy(), max_steps=np.array(4000), global_step=np.array(1700))
    # full gradients for a spread of small tensors; for every tensor its L2 norm and its projection on a fixed pattern
    keep = ("input_blocks.0.0.weight", "input_blocks.1.0.in_layers.0.weight", "input_blocks.1.0.emb_layers.1.bias",
            "input_blocks.2.0.op.bias", "middle_block.1.norm.bias", "middle_block.1.qkv.bias", "middle_block.1.proj_out.bias",
            "output_blocks.1.0.skip_connection.bias", "output_blocks.2.0.out_layers.3.bias", "out.0.weight", "out.2.weight",
            "time_mlp.0.bias", "cond_mlp.2.bias")
    names, norms, projs = [], [], []
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        if n in keep:
            fx["g:" + n] = p.grad.numpy()
        gflat = p.grad.reshape(-1).double()
        pat = torch.cos(torch.arange(gflat.numel(), dtype=torch.float64) * 0.37 + 0.1)
        names.append(n); norms.append(float(gflat.norm())); projs.append(float((gflat * pat).sum()))
    fx["gnames"] = np.array(names); fx["gnorm"] = np.array(norms); fx["gproj"] = np.array(projs)
    np.savez_compressed(os.path.join(mg.OUT, "micro_cm_step.npz"), **fx)
    print("loss", float(loss.detach()), "timesteps", captured["timesteps"].tolist(), "grid", len(captured["pdf"]) + 1,
          os.path.getsize(os.path.join(mg.OUT, "micro_cm_step.npz")))


if __name__ == "__main__":
    main()
