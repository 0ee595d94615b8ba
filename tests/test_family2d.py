"""The dims=2 model family (second half): tqdne_amd/family2d.py runs the reference's
generate_waveforms.py models (architectures.py:40-79) on stock PyTorch operators.  Pinned here against outputs of the
reference itself (tests/golden/micro_2d.npz, written by tools/make_2d_goldens.py): plain torch arithmetic on both sides, so the
tolerance is rounding order only.  The 1-D hot path is untouched by that family -- the last test checks that it still refuses
CPU tensors instead of falling back."""

import warnings

import numpy as np
import pytest
import torch

from conftest import cfg_of, load_golden, rel_err

TOL = 2e-5


@pytest.fixture(scope="module")
def fx():
    sd, d = load_golden("micro_2d.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from tqdne_amd import LightningAutoencoder, LightningEDM
        opt = {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}
        edm = LightningEDM(cfg_of(d, "unet_cfg"), opt, num_sampling_steps=4)
        edm.unet.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("unet.")})
        ae = LightningAutoencoder(cfg_of(d, "enc_cfg"), cfg_of(d, "dec_cfg"), opt)
        ae.load_state_dict({k[3:]: v for k, v in sd.items() if k.startswith("ae.")})
    return edm.eval(), ae.eval(), {k: torch.from_numpy(np.asarray(v)) for k, v in d.items() if not k.endswith("_cfg")}


def test_state_dict_layout_matches_the_reference():
    sd, d = load_golden("micro_2d.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from tqdne_amd import UNetModel
        net = UNetModel(**cfg_of(d, "unet_cfg"))
    ref = {k[5:]: tuple(v.shape) for k, v in sd.items() if k.startswith("unet.")}
    assert list(ref) == list(net.state_dict()) and all(tuple(v.shape) == ref[k] for k, v in net.state_dict().items())
    assert net.input_blocks[0][0].weight.dim() == 4   # Conv2d kernels


def test_unet_forward(fx):
    edm, _, d = fx
    with torch.no_grad():
        y = edm.unet(d["unet:x"], d["unet:t"], d["unet:cond"])
    assert rel_err(y, d["unet:y"]) < TOL


def test_preconditioned_denoiser(fx):
    edm, _, d = fx
    cond = d["unet:cond"]
    for s in (0.002, 0.5, 80.0):
        with torch.no_grad():
            y = edm(d[f"edm:denoise:{s}:x"], torch.full((2,), s), None, cond)
        assert rel_err(y, d[f"edm:denoise:{s}:y"]) < TOL, s


def test_training_loss_and_gradients_through_autograd(fx):
    edm, _, d = fx
    edm.train()
    edm.zero_grad()
    loss = edm.step_with_noise(d["edm:signal"], d["edm:step:eps"], d["edm:step:noise"], cond=d["unet:cond"])
    loss.backward()
    edm.eval()
    assert float(loss.detach()) == pytest.approx(float(d["edm:step:loss"]), rel=1e-5)
    for k in d:
        if k.startswith("edm:step:grad:"):
            g = edm.get_parameter(k[len("edm:step:grad:"):]).grad
            assert rel_err(g, d[k]) < 1e-4, k
    edm.zero_grad()


def test_heun_samplers(fx):
    from tqdne_amd import family2d
    edm, _, d = fx
    cond = d["unet:cond"]
    edm.num_sampling_steps = 4
    sig = edm.edm.sampling_sigmas(4)
    with torch.no_grad():
        out = family2d.heun_sample(edm, d["edm:sample:start"] * sig[0], sig, None, cond).float()
    assert rel_err(out, d["edm:sample:out"]) < TOL
    # the same through sample(): the start state is the first draw of the global generator
    torch.manual_seed(21)
    assert torch.equal(edm.sample((2, 4, 16, 24), cond=cond), out)
    edm.num_sampling_steps = 3
    sig = edm.edm.sampling_sigmas(3)
    draws = iter(d["edm:stoch:churn"])
    with torch.no_grad():
        out = family2d.heun_sample(edm, d["edm:stoch:start"] * sig[0], sig, None, cond, churn=lambda x: next(draws)).float()
    assert rel_err(out, d["edm:stoch:out"]) < TOL
    edm.num_sampling_steps = 4


def test_autoencoder_and_latent_pipeline(fx):
    from tqdne_amd import family2d
    edm, ae, d = fx
    with torch.no_grad():
        z, mean, log_std = ae._encode(d["ae:x"], unit_noise=d["ae:eps"])
        recon = ae.decode(z)
    for got, key in ((z, "ae:z"), (mean, "ae:mean"), (log_std, "ae:log_std"), (recon, "ae:recon")):
        assert rel_err(got, d[key]) < TOL, key
    torch.manual_seed(32)
    ae.train()
    loss = ae.step({"signal": d["ae:x"]})
    ae.eval()
    assert float(loss.detach()) == pytest.approx(float(d["ae:step:loss"]), rel=1e-5)
    # latent EDM: start state in the latent shape, Heun in the latent, decode (edm.py:146-169)
    edm.autoencoder, edm.num_sampling_steps = ae, 3
    try:
        sig = edm.edm.sampling_sigmas(3)
        with torch.no_grad():
            lat = family2d.heun_sample(edm, d["latent:start"] * sig[0], sig, None, d["unet:cond"]).float()
            out = ae.decode(lat)
        assert rel_err(out, d["latent:out"]) < TOL
        assert edm.sample((2, 3, 32, 48), cond=d["unet:cond"]).shape == (2, 3, 32, 48)
    finally:
        edm.autoencoder, edm.num_sampling_steps = None, 4


def test_checkpoint_round_trip(fx, tmp_path):
    """generate_waveforms.py:118-124 loads both models with load_from_checkpoint"""
    from tqdne_amd import LightningAutoencoder, LightningEDM, checkpoint
    edm, ae, d = fx
    f_edm, f_ae = str(tmp_path / "edm2d.ckpt"), str(tmp_path / "ae2d.ckpt")
    edm.autoencoder = ae          # (a latent EDM is saved with its frozen autoencoder inside, as Lightning does)
    try:
        checkpoint.save_checkpoint(edm, f_edm)
    finally:
        edm.autoencoder = None
    checkpoint.save_checkpoint(ae, f_ae)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ae2 = LightningAutoencoder.load_from_checkpoint(f_ae).eval()
        edm2 = LightningEDM.load_from_checkpoint(f_edm, autoencoder=ae2).eval()
    assert edm2.unet.dims == 2 and edm2.autoencoder is ae2
    with torch.no_grad():
        assert torch.equal(edm2.unet(d["unet:x"], d["unet:t"], d["unet:cond"]), edm.unet(d["unet:x"], d["unet:t"], d["unet:cond"]))
        assert torch.equal(ae2.decode(d["ae:z"]), ae.decode(d["ae:z"]))


def test_the_1d_path_has_no_such_fallback():
    from tqdne_amd import UNetModel, tiny_1d_unet_config
    net = UNetModel(**tiny_1d_unet_config()).eval()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64), torch.zeros(1))
