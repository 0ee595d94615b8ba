"""MovingAverageEnvelope (tqdne/representation.py:41-60): oracle vs the reference's own outputs (golden),
HIP vs oracle.  Tolerances: oracle 1e-12 (float64, summation order only); HIP 2e-6 relative (fp32 outputs of a float64
computation) -- the quotient x / (env + eps) amplifies nothing, so fp32 rounding of the result is the whole error."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _z():
    return np.load(os.path.join(GOLDEN, "envelope.npz"))


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def test_oracle_matches_reference_outputs():
    from oracle import representation as R

    z = _z()
    for tag in "abc":
        assert _rel(R.get_representation(z[f"{tag}:x"]), z[f"{tag}:repr"]) < 1e-12
        assert _rel(R.invert_representation(z[f"{tag}:repr"]), z[f"{tag}:inv"]) < 1e-12
        assert _rel(R.invert_representation(z[f"{tag}:repr"].astype(np.float32)), z[f"{tag}:inv_of_f32"]) < 1e-12
    assert _rel(R.get_representation(z["w32:x"], 32, 1e-5, 1e-4), z["w32:repr"]) < 1e-12
    with pytest.raises(ValueError):
        R.get_representation(np.zeros((3, 100), np.float32))  # shorter than the window: the reference's output changes length


@pytest.mark.gpu
def test_hip_envelope_matches_golden_and_oracle():
    from oracle import representation as R
    from tqdne_amd.representation import MovingAverageEnvelope

    z = _z()
    rep = MovingAverageEnvelope()
    for tag in "abc":
        r = rep.get_representation(z[f"{tag}:x"])  # numpy in, numpy out
        assert isinstance(r, np.ndarray) and r.shape == z[f"{tag}:repr"].shape
        assert _rel(r, z[f"{tag}:repr"]) < 2e-6
        inv = rep.invert_representation(z[f"{tag}:repr"].astype(np.float32))
        assert _rel(inv, z[f"{tag}:inv_of_f32"]) < 2e-6
    r = MovingAverageEnvelope(window_size=32, log_eps=1e-5, eps=1e-4).get_representation(z["w32:x"])
    assert _rel(r, z["w32:repr"]) < 2e-6
    with pytest.raises(ValueError):
        rep.get_representation(np.zeros((3, 100), np.float32))
    # the experiments' shape (config.py:61-67: 3 x 4064), a batch, ragged tile edge; GPU tensors in, GPU tensors out
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, 3, 4064, generator=g) * torch.linspace(0.0, 3.0, 4064)
    xd = x.to("cuda:0")
    rd = rep.get_representation(xd)
    assert rd.is_cuda and rd.shape == (64, 6, 4064)
    ref = R.get_representation(x.numpy())
    assert _rel(rd.cpu().numpy(), ref) < 2e-6
    back = rep.invert_representation(rd)
    # round trip: scaled * (env + eps) with env recovered from its fp32 log -> |x| * 1e-6-ish relative
    assert float((back.cpu() - x).abs().max()) < 2e-5 * float(x.abs().max())
