"""Checkpoint compatibility: a Lightning-format .ckpt written by the reference's own classes
(tools/make_ckpt_golden.py) loads into the drop-in module, with and without the EMA weights, and reproduces the
reference's outputs; a file written here uses the reference's class paths."""

import io
import os
import zipfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

CKPT = os.path.join(GOLDEN, "nano_edm.ckpt")


def _expected():
    z = np.load(os.path.join(GOLDEN, "nano_edm_expected.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _oracle_forward(ckpt, sd, e):
    from oracle import edm as oe

    hp = ckpt["hyper_parameters"]
    c = hp["edm"]
    p = oe.EDMParams(sigma_min=c.sigma_min, sigma_max=c.sigma_max, rho=c.rho, sigma_data=c.sigma_data, P_mean=c.P_mean,
                     P_std=c.P_std, S_churn=c.S_churn, S_min=c.S_min, S_max=c.S_max, S_noise=c.S_noise)
    net = oe.make_net(sd, dict(hp["unet_config"]), prefix="unet.")
    with torch.no_grad():
        return oe.denoise(p, net, e["x"], e["sigma"], cond=e["cond"])


def test_reference_checkpoint_loads_without_the_reference():
    from tqdne_amd import checkpoint
    from tqdne_amd.edm import EDM

    ckpt = checkpoint.load_checkpoint(CKPT)
    assert {"state_dict", "hyper_parameters", "ema_state", "optimizer_states", "lr_schedulers"} <= set(ckpt)
    edm = ckpt["hyper_parameters"]["edm"]
    assert type(edm) is EDM and edm.sigma_max == 60.0 and edm.sigma_min == 0.002  # pickled instance state + class defaults
    assert ckpt["epoch"] == 3 and ckpt["global_step"] == 42


def test_checkpoint_weights_reproduce_reference_output_in_oracle():
    """Pins the state_dict schema: the oracle run on the loaded tensors equals the reference module's own output (<= 1e-6)."""
    from tqdne_amd import checkpoint

    ckpt, e = checkpoint.load_checkpoint(CKPT), _expected()
    assert rel_err(_oracle_forward(ckpt, ckpt["state_dict"], e), e["y"]) < 1e-6
    sd_ema = dict(ckpt["state_dict"])
    sd_ema.update(ckpt["ema_state"])
    assert rel_err(_oracle_forward(ckpt, sd_ema, e), e["y_ema"]) < 1e-6


def test_module_from_checkpoint_has_reference_state(tmp_path):
    from tqdne_amd import checkpoint
    from tqdne_amd.edm import LightningEDM

    m = LightningEDM.load_from_checkpoint(CKPT)
    ckpt = checkpoint.load_checkpoint(CKPT)
    sd = m.state_dict()
    assert list(sd) == list(ckpt["state_dict"])
    assert all(torch.equal(sd[k], v) for k, v in ckpt["state_dict"].items())
    assert m.num_sampling_steps == 6 and m.edm.sigma_max == 60.0 and m.optimizer_params["max_steps"] == 100
    m_ema = LightningEDM.load_from_checkpoint(CKPT, ema=True)
    assert all(torch.equal(m_ema.state_dict()[k], v) for k, v in ckpt["ema_state"].items())
    # optimizer / scheduler states restore into the drop-in's own optimizer
    opt = m.configure_optimizers()
    opt["optimizer"].load_state_dict(ckpt["optimizer_states"][0])
    opt["lr_scheduler"]["scheduler"].load_state_dict(ckpt["lr_schedulers"][0])

    # write side: same layout, classes pickled under the reference's module path (no tqdne_amd name in the pickle)
    out = tmp_path / "out.ckpt"
    checkpoint.save_checkpoint(m, out, ema_state=ckpt["ema_state"], optimizer=opt["optimizer"],
                               lr_scheduler=opt["lr_scheduler"]["scheduler"], epoch=4, global_step=50)
    z = zipfile.ZipFile(out)
    pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"tqdne.edm" in pkl and b"tqdne_amd" not in pkl
    back = checkpoint.load_checkpoint(out)
    assert back["global_step"] == 50 and back["hyper_parameters"]["edm"].sigma_max == 60.0
    assert all(torch.equal(back["state_dict"][k], v) for k, v in ckpt["state_dict"].items())


@pytest.mark.gpu
def test_hip_module_from_checkpoint_matches_reference_output():
    """Tolerance: the north-star bar, 1e-3 relative (measured ~1e-5)."""
    from tqdne_amd.edm import LightningEDM

    e = _expected()
    dev = torch.device("cuda:0")
    for ema, key in ((False, "y"), (True, "y_ema")):
        m = LightningEDM.load_from_checkpoint(CKPT, ema=ema).to(dev).eval()
        with torch.no_grad():
            y = m(e["x"].to(dev), e["sigma"].to(dev), cond=e["cond"].to(dev))
        assert rel_err(y.cpu(), e[key]) < 1e-3


def test_checkpoint_with_a_foreign_global_is_refused(tmp_path):
    """a .ckpt is a pickle: anything outside the allow-list (here os.system) must not be resolved, let alone called"""
    import pickle

    from tqdne_amd.checkpoint import load_checkpoint

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    path = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {}, "hyper_parameters": {"x": Evil()}}, path)
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        load_checkpoint(path)
    assert not (tmp_path / "pwned").exists()


def test_checkpoint_cannot_reach_an_unrestricted_unpickler(tmp_path):
    """torch.storage._load_from_bytes is torch.load(weights_only=False) on a bytes argument: a payload nested through it
    would bypass the allow-list, so the function itself is not on it."""
    import pickle

    from tqdne_amd.checkpoint import _SAFE_GLOBALS, load_checkpoint

    assert ("torch.storage", "_load_from_bytes") not in _SAFE_GLOBALS
    marker = tmp_path / "pwned2"

    class Inner:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % marker,))

    inner = io.BytesIO()
    torch.save({"x": Inner()}, inner)  # a complete, unrestricted torch pickle

    class Outer:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    path = tmp_path / "nested.ckpt"
    torch.save({"state_dict": {}, "hyper_parameters": {"x": Outer()}}, path)
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        load_checkpoint(path)
    assert not marker.exists()
