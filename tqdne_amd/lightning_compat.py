"""LightningModule surface for the drop-in modules.

If ``pytorch_lightning`` is importable the real ``pl.LightningModule`` is used (so ``pl.Trainer.fit`` drives the
module exactly as ``experiments/train_1d_edm.py:44-70`` does).  This image has no Lightning, so a minimal base with
the attributes the hot path touches (reference tqdne/edm.py:103,126,138; generate_waveforms.py:122-124) stands in:
``save_hyperparameters``, ``hparams``, ``device``, ``dtype``, ``log``, ``load_from_checkpoint``.  It owns no
arithmetic.
"""

from __future__ import annotations

import inspect

import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ImportError and friends
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = {}
            self._logged = {}

        def save_hyperparameters(self, *args, ignore=(), frame=None):
            frame = frame or inspect.currentframe().f_back
            ignore = (ignore,) if isinstance(ignore, str) else tuple(ignore)
            init = type(self).__init__
            names = [n for n in inspect.signature(init).parameters if n != "self"]
            self.hparams = {n: frame.f_locals[n] for n in names if n in frame.f_locals and n not in ignore}

        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return next(self.parameters()).dtype

        def log(self, name, value, **kwargs):
            self._logged[name] = value

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, ema=False, **kwargs):
            """Lightning's classmethod of the same name (experiments/generate.py:114-120): rebuild the module from the saved
            constructor kwargs (overridden by ``kwargs``, e.g. ``autoencoder=``) and load ``state_dict``.  ``ema=True``
            additionally loads the EMA weights saved by the reference's EMA callback (tqdne/ema.py)."""
            from .checkpoint import apply_ema, load_checkpoint

            ckpt = load_checkpoint(checkpoint_path, map_location=map_location or "cpu")
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            if ema:
                apply_ema(model, ckpt)
            return model
