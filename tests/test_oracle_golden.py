"""Pin the CPU oracle against outputs of the reference itself (tests/golden, made by
tools/make_goldens.py).  Tolerance: 1e-6 relative (same ATen CPU kernels, fp32)."""

import numpy as np
import pytest
import torch

from conftest import cfg_of, load_golden, rel_err
from oracle import autoencoder as AE
from oracle import consistency as CM
from oracle import edm as E
from oracle import unet as U

TOL = 1e-6


@pytest.mark.parametrize("T", [256, 248])
def test_unet_forward_matches_reference(golden_unet, T):
    sd, d = golden_unet
    cfg = cfg_of(d)
    taps = {}
    with torch.no_grad():
        y = U.unet_forward(sd, cfg, torch.from_numpy(d[f"T{T}:x"]), torch.from_numpy(d[f"T{T}:t"]),
                           torch.from_numpy(d[f"T{T}:cond"]), taps=taps)
    assert rel_err(y, d[f"T{T}:y"]) < TOL
    n = 0
    for k in d:
        if k.startswith(f"T{T}:tap:"):
            assert rel_err(taps[k.split(":", 2)[2]][:1], d[k]) < TOL, k
            n += 1
    assert n >= 6


def test_layout_covers_every_weight(golden_unet):
    sd, d = golden_unet
    cfg = cfg_of(d)
    inputs, middle, outputs, ch = U.unet_layout(cfg)
    prefixes = [l[1] for blk in inputs + [middle] + outputs for l in blk]
    for k in sd:
        if k.split(".")[0] in ("input_blocks", "middle_block", "output_blocks"):
            assert any(k.startswith(p + ".") for p in prefixes), k
    # concat GroupNorm groups that straddle the two sources are present in the micro config
    assert any(l[0] == "res" and l[2] == 96 for blk in outputs for l in blk)


def _edm_fixture():
    sd, _ = load_golden("micro_unet.npz")
    _, d = load_golden("micro_edm.npz")
    sd = {"unet." + k: v for k, v in sd.items()}
    return sd, d, cfg_of(d)


def test_edm_scalars_and_schedule():
    _, d, _ = _edm_fixture()
    p = E.EDMParams()
    s = E.sampling_sigmas(p, 18)
    assert s.dtype == torch.float32 and s.shape == (19,)
    assert np.array_equal(s.numpy(), d["sigmas18"])
    assert float(s[0]) == pytest.approx(80.0, rel=1e-6) and float(s[-1]) == 0.0


@pytest.mark.parametrize("sigma", [0.002, 0.5, 80.0])
def test_edm_denoise(sigma):
    sd, d, cfg = _edm_fixture()
    net = E.make_net(sd, cfg)
    x = torch.from_numpy(d[f"denoise:{sigma}:x"])
    with torch.no_grad():
        y = E.denoise(E.EDMParams(), net, x, torch.full((x.shape[0],), sigma), None, torch.from_numpy(d["cond"]))
    assert rel_err(y, d[f"denoise:{sigma}:y"]) < TOL


def test_edm_loss_and_grads():
    sd, d, cfg = _edm_fixture()
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and k != "unet.time_embed.W") for k, v in sd.items()}
    net = E.make_net(sd, cfg)
    loss = E.loss_step(E.EDMParams(), net, torch.from_numpy(d["signal"]), torch.from_numpy(d["step:eps"]),
                       torch.from_numpy(d["step:noise"]), cond=torch.from_numpy(d["cond"]))
    assert rel_err(loss, d["step:loss"]) < TOL
    loss.backward()
    n = 0
    for k in d:
        if k.startswith("step:grad:"):
            g = sd[k[len("step:grad:"):]].grad
            assert rel_err(g, d[k]) < 2e-5, k  # backward reductions re-associate; still fp32 ATen
            n += 1
    assert n >= 15


def test_edm_sampler_deterministic():
    sd, d, cfg = _edm_fixture()
    net = E.make_net(sd, cfg)
    trace = {}
    with torch.no_grad():
        out = E.sample_deterministic(E.EDMParams(), net, torch.from_numpy(d["sample:start"]), 18,
                                     cond=torch.from_numpy(d["cond"]), trace=trace)
    assert int(d["sample:nfe"]) == 35
    assert rel_err(trace[1], d["sample:state1"]) < TOL
    assert rel_err(trace[9], d["sample:state9"]) < 1e-5
    assert rel_err(out.float(), d["sample:out"]) < 1e-5


def test_edm_sampler_stochastic():
    sd, d, cfg = _edm_fixture()
    net = E.make_net(sd, cfg)
    with torch.no_grad():
        out = E.sample_stochastic(E.EDMParams(), net, torch.from_numpy(d["stoch:start"]),
                                  [torch.from_numpy(c) for c in d["stoch:churn"]], 6, cond=torch.from_numpy(d["cond"]))
    assert rel_err(out.float(), d["stoch:out"]) < 1e-5


def test_consistency_sampler():
    sd, _ = load_golden("micro_unet.npz")
    _, d = load_golden("micro_cm.npz")
    cfg = cfg_of(d)

    def net(x, t, c):
        return U.unet_forward(sd, cfg, x, t, c)

    cond = torch.from_numpy(d["cond"])
    with torch.no_grad():
        y1 = CM.sample(net, torch.from_numpy(d["start"]), cond=cond)
        y2 = CM.sample(net, torch.from_numpy(d["start"]), [1.0], [torch.from_numpy(d["uniform"])], cond=cond)
    assert rel_err(y1, d["one_step"]) < TOL
    assert rel_err(y2, d["refined"]) < 1e-5


def test_autoencoder_encode_decode():
    sd, d = load_golden("micro_ae.npz")
    enc_cfg, dec_cfg = cfg_of(d, "enc_cfg"), cfg_of(d, "dec_cfg")
    with torch.no_grad():
        z, mean, log_std = AE.encode(sd, enc_cfg, torch.from_numpy(d["x"]), torch.from_numpy(d["eps"]))
        xr = AE.decode(sd, dec_cfg, z)
    assert rel_err(mean, d["mean"]) < TOL and rel_err(log_std, d["log_std"]) < TOL
    assert rel_err(z, d["z"]) < TOL
    assert rel_err(xr, d["recon"]) < TOL


def test_ict_training_step_vs_reference_step():
    """consistency_model.py:115-176 restated (schedule, discretised lognormal, pseudo-Huber loss): loss, pdf and the gradient
    norms of every parameter against the reference's own step (tools/make_cm_step_golden.py)"""
    import os
    import numpy as np
    from conftest import GOLDEN
    from oracle import consistency as OC, unet as OU
    sd, d = load_golden("micro_unet.npz")
    s = np.load(os.path.join(GOLDEN, "micro_cm_step.npz"))
    cfg = cfg_of(d)
    params = {k: v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    net = lambda x, t, c: OU.unet_forward(params, cfg, x, t, c)
    sig = OC.ict_schedule(int(s["global_step"]), int(s["max_steps"]))
    assert len(sig) == len(s["pdf"]) + 1
    assert rel_err(OC.ict_timestep_pdf(sig), s["pdf"]) < 1e-6
    loss = OC.ict_loss(net, torch.from_numpy(s["sample"]), sig, torch.from_numpy(s["timesteps"]), torch.from_numpy(s["eps"]),
                       torch.from_numpy(s["cond"]))
    assert rel_err(loss.detach(), s["loss"]) < 1e-6
    loss.backward()
    for n, gn in zip(s["gnames"], s["gnorm"]):
        assert abs(float(params[str(n)].grad.double().norm()) - gn) <= 1e-4 * max(gn, 1e-9), n
