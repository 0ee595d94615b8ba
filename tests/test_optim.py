"""Fused Adam + EMA launch against torch.optim.Adam + torch._foreach_lerp_ on the CPU, the update the
reference configures (tqdne/edm.py:240-251) and its EMA callback applies (tqdne/ema.py:24-28).  Tolerance 1e-6 relative:
the same fp32 formula, differences are fma contraction only."""

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 5), (3,), (32,), (256, 768), (5, 7, 3), (4097,), (1,), (130, 33)]


def _make(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) * 0.3 for s in SHAPES]


def test_fused_adam_matches_torch_adam_and_ema_lerp():
    from tqdne_amd.optim import FusedAdamEMA

    dev = torch.device("cuda:0")
    ref = [torch.nn.Parameter(t.clone()) for t in _make()]
    hip = [torch.nn.Parameter(t.clone().to(dev)) for t in _make()]
    opt_ref = torch.optim.Adam(ref, lr=3e-3)
    sch_ref = torch.optim.lr_scheduler.CosineAnnealingLR(opt_ref, T_max=10, eta_min=1e-5)
    opt = FusedAdamEMA([(f"p{i}", p) for i, p in enumerate(hip)], lr=3e-3, ema_decay=0.9)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-5)
    ema_ref = [p.detach().clone() for p in ref]
    g = torch.Generator().manual_seed(7)
    for step in range(4):
        for pr, ph in zip(ref, hip):
            gr = torch.randn(pr.shape, generator=g) * (10.0 ** (step - 2))
            pr.grad = gr.clone()
            ph.grad = (gr * 4.0).to(dev)  # the launch folds the 1 / world_size of the gradient mean
        opt_ref.step()
        torch._foreach_lerp_(tuple(ema_ref), tuple(p.detach() for p in ref), 1 - 0.9)
        sch_ref.step()
        opt.step(grad_scale=0.25)
        sch.step()
    for i, (pr, ph) in enumerate(zip(ref, hip)):
        assert rel_err(ph.detach().cpu(), pr.detach()) < 1e-6, i
        assert rel_err(opt.state[ph]["exp_avg"].cpu(), opt_ref.state[pr]["exp_avg"]) < 1e-6
        assert rel_err(opt.state[ph]["exp_avg_sq"].cpu(), opt_ref.state[pr]["exp_avg_sq"]) < 1e-6
    for (name, e), er in zip(opt.ema_state().items(), ema_ref):
        assert rel_err(e.cpu(), er) < 1e-6, name


def test_fused_adam_resumes_from_torch_adam_state():
    """optimizer_states of a reference checkpoint (torch Adam format) continue identically in the fused optimizer."""
    from tqdne_amd.optim import FusedAdamEMA

    dev = torch.device("cuda:0")
    ref = [torch.nn.Parameter(t.clone()) for t in _make(1)]
    opt_ref = torch.optim.Adam(ref, lr=1e-3)
    g = torch.Generator().manual_seed(3)
    grads = [[torch.randn(p.shape, generator=g) for p in ref] for _ in range(4)]
    for step in range(2):
        for p, gr in zip(ref, grads[step]):
            p.grad = gr.clone()
        opt_ref.step()
    hip = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    opt = FusedAdamEMA([(f"p{i}", p) for i, p in enumerate(hip)], lr=1e-3)
    opt.load_state_dict(opt_ref.state_dict())
    for step in range(2, 4):
        for pr, ph, gr in zip(ref, hip, grads[step]):
            pr.grad = gr.clone()
            ph.grad = gr.to(dev)
        opt_ref.step()
        opt.step()
    for pr, ph in zip(ref, hip):
        assert rel_err(ph.detach().cpu(), pr.detach()) < 1e-6
    sd = opt.state_dict()  # torch format out as well
    assert float(sd["state"][0]["step"]) == 4.0 and sd["state"][0]["exp_avg"].shape == ref[0].shape


def test_fused_adamw_matches_torch_adamw():
    """the autoencoder's optimizer (autoencoder.py:93-95): AdamW, weight_decay 1e-4 (here larger, to be visible)"""
    from tqdne_amd.optim import FusedAdamEMA

    dev = torch.device("cuda:0")
    ref = [torch.nn.Parameter(t.clone()) for t in _make(2)]
    hip = [torch.nn.Parameter(t.clone().to(dev)) for t in _make(2)]
    opt_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.05)
    opt = FusedAdamEMA([(f"p{i}", p) for i, p in enumerate(hip)], lr=1e-2, weight_decay=0.05)
    g = torch.Generator().manual_seed(4)
    for step in range(3):
        for pr, ph in zip(ref, hip):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad, ph.grad = gr.clone(), gr.to(dev)
        opt_ref.step()
        opt.step()
    for pr, ph in zip(ref, hip):
        assert rel_err(ph.detach().cpu(), pr.detach()) < 1e-6
