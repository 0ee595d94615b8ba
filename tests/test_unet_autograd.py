"""``UNetModel`` and ``LightningEDM.forward`` as ordinary differentiable modules (reference tqdne/unet.py:360-398, edm.py:105-113: plain
autograd with respect to the parameters AND the input).  Round-5 verdict, missing item 3: ``loss = f(unet(x, t, c)); loss.backward()``
used to fail -- only the loss wrappers were differentiable.  Every gradient is held against torch autograd through the CPU oracle."""

import pytest
import torch

from conftest import cfg_of, grad_err, load_golden, rel_err
from test_hip_unet import dev, perturbed_state

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _model(which):
    from tqdne_amd import UNetModel, paper_1d_unet_config
    if which == "micro":
        sd, d = load_golden("micro_unet.npz")
        cfg = dict(cfg_of(d), dropout=0.0)
        m = UNetModel(**cfg)
        m.load_state_dict(sd)
        return m, sd, cfg, (2, 256)
    cfg = dict(paper_1d_unet_config(), dropout=0.0)
    torch.manual_seed(0)
    m = UNetModel(**cfg)
    sd = perturbed_state(m, 29)
    m.load_state_dict(sd)
    return m, sd, cfg, (2, 4096)


def _compare_param_grads(named_params, ref_params, prefix=""):
    gmax = max(float(v.grad.abs().max()) for v in ref_params.values() if v.grad is not None)
    worst, wname = 0.0, ""
    n = 0
    for name, p in named_params:
        if not p.requires_grad:
            assert p.grad is None
            continue
        assert p.grad is not None, name
        e = grad_err(p.grad, ref_params[prefix + name].grad, gmax, name)
        n += 1
        if e > worst:
            worst, wname = e, name
    return worst, wname, n


@pytest.mark.parametrize("which", ["micro", "paper"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_unet_module_backward_parameters_and_input(which, mode):
    """unet(x, t, c).square().mean().backward(): all parameter gradients and x.grad vs oracle autograd (dropout 0, so that train and
    eval mode both compare); a second forward + backward accumulates into .grad like any module."""
    from oracle import unet as OU
    m, sd, cfg, (B, T) = _model(which)
    m = m.to(dev())
    m.train(mode == "train")
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, 3, T, generator=g)
    t = torch.randn(B, generator=g) * 0.5
    c = torch.randn(B, 5, generator=g)
    xg = x.to(dev()).requires_grad_(True)
    y = m(xg, t.to(dev()), c.to(dev()))
    assert y.requires_grad and y.grad_fn is not None
    y.square().mean().backward()
    ref = {k: v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yo = OU.unet_forward(ref, cfg, xr, t, c)
    yo.square().mean().backward()
    assert rel_err(y.detach().cpu(), yo.detach()) < TOL
    worst, wname, n = _compare_param_grads(m.named_parameters(), ref)
    ex = rel_err(xg.grad.cpu(), xr.grad)
    print(f"{which} / {mode}: {n} parameter gradients, worst {worst:.2e} ({wname}); d/dx {ex:.2e}")
    assert worst < TOL and ex < TOL and n == len(sd) - 1
    # .grad accumulates over a second call, as with any nn.Module
    g1 = {n_: p.grad.clone() for n_, p in m.named_parameters() if p.grad is not None}
    gmax = max(float(v.abs().max()) for v in g1.values())
    m(x.to(dev()), t.to(dev()), c.to(dev())).square().mean().backward()
    for n_, p in m.named_parameters():
        if p.grad is not None:   # (tensors whose true gradient is zero hold rounding noise only: measured against the largest gradient)
            assert float((p.grad - 2 * g1[n_]).abs().max()) < 1e-4 * max(float(g1[n_].abs().max()), 1e-3 * gmax), n_


def test_unet_forward_without_grad_is_not_recorded_and_non_leaf_inputs_work():
    from oracle import unet as OU
    m, sd, cfg, (B, T) = _model("micro")
    m = m.to(dev()).train()
    g = torch.Generator().manual_seed(43)
    x, t, c = torch.randn(B, 3, T, generator=g), torch.rand(B, generator=g), torch.randn(B, 5, generator=g)
    with torch.no_grad():
        assert not m(x.to(dev()), t.to(dev()), c.to(dev())).requires_grad
    # the input is itself the output of differentiable torch ops (a learned pre-scale): the gradient flows through
    s = torch.tensor(0.7, device=dev(), requires_grad=True)
    (m(x.to(dev()) * s, t.to(dev()), c.to(dev())) * 3.0).sum().backward()
    ref = {k: v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    sr = torch.tensor(0.7, requires_grad=True)
    (OU.unet_forward(ref, cfg, x * sr, t, c) * 3.0).sum().backward()
    assert abs(float(s.grad) - float(sr.grad)) < TOL * abs(float(sr.grad))
    with pytest.raises(NotImplementedError, match="cond"):
        m(x.to(dev()), t.to(dev()), c.to(dev()).requires_grad_(True))
    # two forwards of one shape, then the backward of the FIRST: loud, not silently wrong
    y1 = m(x.to(dev()), t.to(dev()), c.to(dev()))
    m(x.to(dev()), t.to(dev()), c.to(dev()))
    with pytest.raises(RuntimeError, match="another forward"):
        y1.sum().backward()


@pytest.mark.parametrize("concat", [False, True])
def test_edm_forward_input_gradient(concat):
    """LightningEDM.forward (edm.py:105-113) with ``sample.requires_grad``: d D / d sample = c_skip + c_out dF/dx_in c_in, with and
    without a concatenated conditioning signal, vs oracle autograd; parameter gradients in the same backward."""
    from oracle import edm as OE
    from tqdne_amd import LightningEDM, tiny_1d_unet_config
    cfg = dict(tiny_1d_unet_config(in_channels=6 if concat else 3, out_channels=3), dropout=0.0)
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
    sd = perturbed_state(edm.unet, 37)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev()).train()
    g = torch.Generator().manual_seed(8)
    B, T = 2, 1024
    x = torch.randn(B, 3, T, generator=g)
    cs = torch.randn(B, 3, T, generator=g) if concat else None
    sigma = torch.tensor([0.3, 7.0])
    G = torch.randn(B, 3, T, generator=g)
    xg = x.to(dev()).requires_grad_(True)
    y = edm(xg, sigma.to(dev()), cond_sample=cs.to(dev()) if concat else None)
    (y * G.to(dev())).sum().backward()
    params = {("unet." + k): v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yo = OE.denoise(OE.EDMParams(), OE.make_net(params, cfg), xr, sigma, cond_sample=cs)
    (yo * G).sum().backward()
    assert rel_err(y.detach().cpu(), yo.detach()) < TOL
    ex = rel_err(xg.grad.cpu(), xr.grad)
    worst, wname, n = _compare_param_grads(edm.unet.named_parameters(), params, prefix="unet.")
    print(f"EDM forward, concat={concat}: d/dx {ex:.2e}; {n} parameter gradients, worst {worst:.2e} ({wname})")
    assert ex < TOL and worst < TOL
