import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    data = {k: z[k] for k in z.files if not k.startswith("w:")}
    return sd, data


def cfg_of(data, key="cfg"):
    return eval(str(data[key]), {"__builtins__": {}}, {"dict": dict})  # repr() of a plain dict of literals


ELEM_RTOL = 1e-3  # SURVEY.md 8c(4), second metric: allclose(rtol = 1e-3, atol = 1e-3 * rms(reference))
ELEM_LOG = []    # (test id, norm-wise error, element-wise error) of every comparison of the session, printed at the end with -s


def elem_err(a, b):
    """Element-wise companion of rel_err: allclose(a, b, rtol=r, atol=r*rms(b)) holds iff  max_i |a_i - b_i| / (|b_i| + rms(b)) <= r."""
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
    return float(((a - b).abs() / (b.abs() + rms)).max())


def rel_err(a, b, elem=True):
    """Norm-wise error max|a-b| / max|b| (returned; every caller holds it against its own bar, at most the north star's 1e-3).
    With ``elem`` (default) the comparison must ALSO satisfy SURVEY 8c(4)'s element-wise criterion
    allclose(rtol=1e-3, atol=1e-3*rms(b)) -- asserted here so that every parity check of the suite carries both metrics.
    ``elem=False`` only where the second tensor is not a reference (e.g. "did the weights move")."""
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    if elem:
        ee = elem_err(a, b)
        ELEM_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], e, ee))
        assert ee <= ELEM_RTOL, (f"element-wise parity: max|a-b| / (|b| + rms(b)) = {ee:.3e} > {ELEM_RTOL:g} "
                                 f"(norm-wise error of the same comparison: {e:.3e})")
    return e


GRAD_LOG = []   # (test id, tensor name, floored error, own-scale error, |ref|max / gmax)
GRAD_OWN_TOL = 1e-3     # bar on a gradient tensor's error against its OWN largest entry (measured <= 4.2e-4, profiles/r03_*) ...
GRAD_OWN_FROM = 1e-5    # ... for every tensor whose largest reference entry is at least this fraction of the largest gradient


def grad_err(g, ref, gmax, name=""):
    """Error of one parameter gradient.  Returned: max|g-ref| / max(max|ref|, 1e-3*gmax) -- the suite's bar (1e-3) applies to it;
    tensors far below the largest gradient are pure rounding noise on both sides in places (a bias feeding a one-channel
    GroupNorm group is exactly zero in exact arithmetic), hence the floor.  So that the floor cannot hide a wrong SMALL
    gradient, the error against the tensor's own scale is asserted too (GRAD_OWN_TOL) wherever the reference is above
    GRAD_OWN_FROM * gmax, and logged for the end-of-session table."""
    g = torch.as_tensor(g).detach().to(torch.float64).cpu()
    ref = torch.as_tensor(ref).detach().to(torch.float64).cpu()
    d = float((g - ref).abs().max())
    rmax = float(ref.abs().max())
    floored = d / max(rmax, 1e-3 * gmax, 1e-300)
    own = d / max(rmax, 1e-300)
    GRAD_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], name, floored, own, rmax / max(gmax, 1e-300)))
    if rmax >= GRAD_OWN_FROM * gmax:
        assert own <= GRAD_OWN_TOL, f"gradient {name}: error {own:.3e} of its own scale (|ref|max = {rmax / gmax:.2e} of the largest gradient)"
    return floored


def pytest_terminal_summary(terminalreporter):
    if GRAD_LOG and os.environ.get("TQDNE_PARITY_LOG", "1") != "0":
        per = {}
        for tid, name, fl, own, frac in GRAD_LOG:
            w = per.setdefault(tid, [0.0, "", 0.0, "", 0.0, 0])
            if fl > w[0]:
                w[0], w[1] = fl, name
            if frac >= GRAD_OWN_FROM and own > w[2]:
                w[2], w[3], w[4] = own, name, frac
            w[5] += 1
        terminalreporter.write_sep("-", "gradient parity per test: worst floored error (tensor); worst own-scale error (tensor, |ref|max/gmax); tensors")
        for tid, w in per.items():
            terminalreporter.write_line(f"{tid}: {w[0]:.2e} ({w[1]}); {w[2]:.2e} ({w[3]}, {w[4]:.1e}); {w[5]}")
    _elem_summary(terminalreporter)


def _elem_summary(terminalreporter):
    if not ELEM_LOG or os.environ.get("TQDNE_PARITY_LOG", "1") == "0":
        return
    worst = {}
    for tid, e, ee in ELEM_LOG:
        w = worst.setdefault(tid, [0.0, 0.0, 0])
        w[0], w[1], w[2] = max(w[0], e), max(w[1], ee), w[2] + 1
    terminalreporter.write_sep("-", "parity metrics per test: worst norm-wise max|a-b|/max|b|, worst element-wise |a-b|/(|b|+rms b), comparisons")
    for tid, (e, ee, n) in worst.items():
        terminalreporter.write_line(f"{tid}: {e:.2e} {ee:.2e} ({n})")


@pytest.fixture(scope="session")
def golden_unet():
    return load_golden("micro_unet.npz")
