"""A12: LightningDDMP (tqdne/diffusion.py).  PARITY UNPINNED (the reference's scheduler lives in diffusers, absent from its lockfile and
from this image): the HIP path is checked against oracle/diffusion.py, an independent float64 restatement of the published DDPM
equations, and the scheduler through algebraic properties of those equations."""

import numpy as np
import pytest
import torch

from conftest import cfg_of, grad_err, load_golden, rel_err

TOL = 1e-3


def test_scheduler_tables_and_step_properties_cpu():
    """host logic only: the coefficient identities of Ho et al. eq. 7 and the 'leading' inference grid"""
    from oracle import diffusion as OD
    from tqdne_amd.diffusion import DDPMScheduler, get_cosine_schedule_with_warmup
    s = DDPMScheduler()
    betas, abar = OD.schedule()
    assert np.allclose(s.alphas_cumprod.double().numpy(), abar, rtol=2e-6)
    assert s.timesteps[0] == 999 and s.timesteps[-1] == 0 and len(s.timesteps) == 1000
    for t in (999, 500, 1):
        s1, inv, c0, ct, sigma = s.step_coefficients(t)
        # with the exact noise and no clipping the predicted x0 is exact, and the posterior mean of x0 = x_t = 0 is 0
        assert abs(s1 ** 2 + 1.0 / inv ** 2 - 1.0) < 1e-12
        # eq. 7: coef_x0 sqrt(abar_t) + coef_xt = sqrt(abar_{t-1})-weighted consistency: E[x_{t-1} | x0] = sqrt(abar_prev) x0
        assert abs(c0 + ct * (1.0 / inv) - np.sqrt(abar[t - 1])) < 1e-6
        assert sigma > 0
    assert s.step_coefficients(0)[4] == 0.0
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=1.0)
    sch = get_cosine_schedule_with_warmup(opt, 10, 110)
    lrs = []
    for _ in range(110):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert lrs[0] == 0.0 and abs(lrs[10] - 1.0) < 1e-12 and abs(lrs[60] - 0.5) < 1e-9 and lrs[-1] < 1e-3


def _module(prediction_type="epsilon", cond_signal_input=False):
    from tqdne_amd import UNetModel
    from tqdne_amd.diffusion import DDPMScheduler, LightningDDMP
    sd, d = load_golden("micro_unet.npz")
    cfg = dict(cfg_of(d), dropout=0.0)
    if cond_signal_input:
        cfg["in_channels"] = 2 * cfg["in_channels"]
    torch.manual_seed(3)
    net = UNetModel(**cfg)
    if not cond_signal_input:
        net.load_state_dict(sd)
    else:
        from test_hip_unet import perturbed_state
        net.load_state_dict(perturbed_state(net, 5))
    m = LightningDDMP(net, DDPMScheduler(), {"learning_rate": 1e-4, "lr_warmup_steps": 5, "n_train": 10, "max_epochs": 2},
                      prediction_type=prediction_type, cond_signal_input=cond_signal_input, cond_input=True)
    return m, cfg, {k: v.clone() for k, v in net.state_dict().items()}


@pytest.mark.gpu
@pytest.mark.parametrize("prediction_type,cond_signal_input", [("epsilon", False), ("sample", True)])
def test_ddpm_step_loss_and_gradients_vs_oracle(prediction_type, cond_signal_input):
    from oracle import diffusion as OD
    from oracle import unet as OU
    dev = torch.device("cuda:0")
    m, cfg, sd = _module(prediction_type, cond_signal_input)
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(11)
    B, T = 3, 256
    x0 = 0.5 * torch.randn(B, 3, T, generator=g)
    cs = torch.randn(B, 3, T, generator=g) if cond_signal_input else None
    cond = torch.randn(B, 5, generator=g)
    noise = torch.randn(B, 3, T, generator=g)
    t = torch.tensor([7, 512, 993])
    batch = {"signal": x0.to(dev), "cond": cond.to(dev)}
    if cs is not None:
        batch["cond_signal"] = cs.to(dev)
    loss = m.step_with_noise(batch, noise.to(dev), t.to(dev))
    loss.backward()
    params = {k: v.clone().requires_grad_(v.is_floating_point() and k != "time_embed.W") for k, v in sd.items()}
    _, abar = OD.schedule()
    lo = OD.loss(lambda x, ts, c: OU.unet_forward(params, cfg, x, ts, c), x0, noise, t.numpy(), abar, prediction_type, cs, cond)
    lo.backward()
    e = rel_err(loss.detach().cpu(), lo.detach())
    gmax = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    worst = 0.0
    for name, p in m.net.named_parameters():
        if p.requires_grad and params[name].grad is not None:
            worst = max(worst, grad_err(p.grad, params[name].grad, gmax, name))
    print(f"DDPM step ({prediction_type}, cond_signal {cond_signal_input}): loss rel err {e:.2e}, worst gradient {worst:.2e}")
    assert e < TOL and worst < TOL


@pytest.mark.gpu
def test_ddpm_ancestral_sampling_vs_oracle():
    from oracle import diffusion as OD
    from oracle import unet as OU
    dev = torch.device("cuda:0")
    m, cfg, sd = _module()
    m = m.to(dev).eval()
    m.noise_scheduler.set_timesteps(10)
    g = torch.Generator().manual_seed(12)
    B, T = 2, 256
    start = torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g)
    noises = [torch.randn(B, 3, T, generator=g) for _ in range(10)]
    out = m.sample((B, 3, T), cond=cond.to(dev), start=start.to(dev), noises=[z.to(dev) for z in noises]).cpu()
    _, abar = OD.schedule()
    with torch.no_grad():
        ref = OD.sample(lambda x, ts, c: OU.unet_forward(sd, cfg, x, ts, c), start.double(), [z.double() for z in noises],
                        m.noise_scheduler.timesteps.tolist(), 100, abar, cond=cond)
    e = rel_err(out, ref)
    print(f"DDPM 10-step ancestral sample vs oracle: {e:.2e}")
    assert e < TOL
    # forward-process property through the kernels: x_t built from x0 and eps, then one exact-eps step with clipping off and no
    # noise (t small) must move towards x0: E[x_{t-1} | x0, x_t] of eq. 7
    s = m.noise_scheduler
    s.config.clip_sample = False
    x0 = 0.3 * torch.randn(B, 3, T, generator=g).to(dev)
    eps = torch.randn(B, 3, T, generator=g).to(dev)
    xt = s.add_noise(x0, eps, torch.tensor([300, 300]))
    prev = s.step(eps, 300, xt, noise=torch.zeros_like(xt)).prev_sample
    _, _, c0, ct, _ = s.step_coefficients(300)
    assert rel_err(prev.cpu(), (c0 * x0 + ct * xt).cpu()) < 1e-5
