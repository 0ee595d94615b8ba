#!/usr/bin/env python3
"""Generate tests/golden/envelope.npz with the reference's own MovingAverageEnvelope (tqdne/representation.py:41-60).
`pathos`, `PIL` and `pytorch_lightning` are imported by the reference's module headers but not used by this class; empty
in-memory stand-ins let the import succeed (they contribute no arithmetic).   Run: python tools/make_envelope_golden.py"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


def main():
    sys.path.insert(0, mg.REF)
    mg.install_lightning_standin()
    for name in ("pathos", "pathos.multiprocessing"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.Pool = None
            sys.modules[name] = m
    try:
        import PIL  # noqa: F401
    except ImportError:
        sys.modules["PIL"] = types.ModuleType("PIL")
    from tqdne.representation import MovingAverageEnvelope

    rng = np.random.default_rng(11)
    fx = {}
    for tag, shape, scale in (("a", (2, 3, 300), 1.0), ("b", (1, 3, 128), 1e-3), ("c", (3, 517), 50.0)):
        t = np.arange(shape[-1])
        x = (rng.standard_normal(shape) * scale * np.exp(-((t - shape[-1] / 3) / (shape[-1] / 5)) ** 2)).astype(np.float32)
        x[..., :7] = 0.0  # silent stretch: env + eps and log(env + log_eps) matter there
        rep = MovingAverageEnvelope()
        r = rep.get_representation(x)
        fx[f"{tag}:x"] = x
        fx[f"{tag}:repr"] = r
        fx[f"{tag}:inv"] = rep.invert_representation(r)
        r32 = r.astype(np.float32)
        fx[f"{tag}:inv_of_f32"] = rep.invert_representation(r32)
    rep = MovingAverageEnvelope(window_size=32, log_eps=1e-5, eps=1e-4)
    x = rng.standard_normal((2, 3, 200)).astype(np.float32)
    fx["w32:x"] = x
    fx["w32:repr"] = rep.get_representation(x)
    np.savez_compressed(os.path.join(mg.OUT, "envelope.npz"), **fx)
    print({k: (v.shape, str(v.dtype)) for k, v in fx.items()})


if __name__ == "__main__":
    main()
