"""Autoencoder training step (autoencoder.py:59-84): MSE + KL loss and every parameter gradient, with the
cond_signal term, against the reference's own step (tests/golden/micro_ae_step.npz, tools/make_ae_step_golden.py).
Oracle <= 1e-6; HIP <= 1e-3 (north-star bar; measured ~1e-5), gradients measured against max(|ref|, 1e-3 * largest gradient)."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, cfg_of, load_golden, rel_err


def _fixture():
    sd, d = load_golden("micro_ae.npz")
    z = np.load(os.path.join(GOLDEN, "micro_ae_step.npz"))
    return sd, d, {k: z[k] for k in z.files}


def test_oracle_ae_step_matches_reference():
    from oracle import autoencoder as OA

    sd, d, s = _fixture()
    enc_cfg, dec_cfg = cfg_of(d, "enc_cfg"), cfg_of(d, "dec_cfg")
    kw = float(s["kl_weight"])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    l0, _, _ = OA.step_loss(params, enc_cfg, dec_cfg, torch.from_numpy(s["x"]), torch.from_numpy(s["eps0"]), kw)
    assert rel_err(l0.detach(), s["loss_signal_only"]) < 1e-6
    l1, _, _ = OA.step_loss(params, enc_cfg, dec_cfg, torch.from_numpy(s["cond_x"]), torch.from_numpy(s["eps1"]), kw)
    loss = l0 + l1
    assert rel_err(loss.detach(), s["loss"]) < 1e-6
    loss.backward()
    gmax = max(float(np.abs(s["g:" + k]).max()) for k in params)
    for k, v in params.items():
        # fp32 autograd on both sides, different summation order; gradients that are zero in exact arithmetic (a conv bias in
        # front of a GroupNorm with one channel per group) are rounding noise of ~1e-10 and differ between hosts: hence the floor
        ref = torch.from_numpy(s["g:" + k])
        assert float((v.grad - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax) < 2e-5, k


@pytest.mark.gpu
def test_hip_ae_step_matches_reference():
    from tqdne_amd import LightningAutoencoder

    sd, d, s = _fixture()
    dev = torch.device("cuda:0")
    ae = LightningAutoencoder(cfg_of(d, "enc_cfg"), cfg_of(d, "dec_cfg"), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0},
                              kl_weight=float(s["kl_weight"]))
    ae.load_state_dict(sd)
    ae = ae.to(dev).eval()  # the golden was taken in eval mode (dropout masks cannot be matched)
    draws = iter([torch.from_numpy(s["eps0"]).to(dev), torch.from_numpy(s["eps1"]).to(dev)])
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: next(draws)
    try:
        loss = ae.step({"signal": torch.from_numpy(s["x"]).to(dev), "cond_signal": torch.from_numpy(s["cond_x"]).to(dev)})
    finally:
        torch.randn_like = orig
    assert rel_err(loss.detach().cpu(), s["loss"]) < 1e-3
    loss.backward()
    gmax = max(float(np.abs(v).max()) for k, v in s.items() if k.startswith("g:"))
    worst, wname = 0.0, ""
    for n, p in ae.named_parameters():
        ref = torch.from_numpy(s["g:" + n])
        e = float((p.grad.cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        if e > worst:
            worst, wname = e, n
    print(f"AE step: loss {float(loss):.6f}; worst gradient rel err {worst:.2e} at {wname}")
    assert worst < 1e-3

    # fused path of the trainer, train mode (dropout active): finite loss, a gradient for every parameter
    ae.train()
    for p in ae.parameters():
        p.grad = None
    total, grads = ae.step_and_backward({"signal": torch.from_numpy(s["x"]).to(dev)})
    assert torch.isfinite(total) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in ae.parameters())


@pytest.mark.gpu
def test_trainer_drives_the_autoencoder():
    """DataParallelTrainer on the autoencoder: fused AdamW (weight decay from configure_optimizers), loss goes down."""
    from tqdne_amd import LightningAutoencoder
    from tqdne_amd.trainer import DataParallelTrainer

    sd, d, s = _fixture()
    dev = torch.device("cuda:0")
    ae = LightningAutoencoder(cfg_of(d, "enc_cfg"), cfg_of(d, "dec_cfg"), {"learning_rate": 2e-3, "max_steps": 50, "eta_min": 0})
    ae.load_state_dict(sd)
    ae = ae.to(dev).train()
    tr = DataParallelTrainer(ae, world_size=1)
    assert tr.fused and tr.optimizer.param_groups[0]["weight_decay"] == 1e-4
    batch = {"signal": torch.from_numpy(s["x"]).to(dev)}
    losses = [float(tr.train_step(batch)) for _ in range(12)]
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0]
