"""Checkpoint compatibility with the reference's Lightning ``.ckpt`` files.

The reference saves with ``pl.callbacks.ModelCheckpoint`` (tqdne/training.py:54-65) and loads with
``LightningEDM.load_from_checkpoint(path, autoencoder=...)`` after ``add_safe_globals([tqdne.edm.EDM])``
(experiments/generate.py:114-120, experiments/evaluate.py:40-48).  Such a file is a ``torch.save``d dict:

    state_dict         parameter name -> tensor (the schema of tqdne_amd.UNetModel is identical, 311 tensors for the paper net)
    hyper_parameters   constructor kwargs; for LightningEDM they include a pickled ``tqdne.edm.EDM`` instance
    ema_state          written by the EMA callback (tqdne/ema.py:50-54): name -> EMA tensor of every trainable parameter
    optimizer_states, lr_schedulers, epoch, global_step, pytorch-lightning_version, ...

``load_checkpoint`` reads such a file without the reference installed: classes pickled under ``tqdne.*`` are resolved to their
``tqdne_amd`` counterparts.  ``save_checkpoint`` writes the same layout (classes are pickled under their ``tqdne.*`` names), so
a file written here loads in the reference and vice versa.
"""

from __future__ import annotations

import importlib
import io
import pickle
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch

# pickled module path of the reference -> module of this package holding the class of the same name
_MODULE_MAP = {
    "tqdne.edm": "tqdne_amd.edm",
    "tqdne.consistency_model": "tqdne_amd.consistency_model",
    "tqdne.autoencoder": "tqdne_amd.autoencoder",
    "tqdne.unet": "tqdne_amd.unet",
}


class _RemapUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        target = _MODULE_MAP.get(module)
        if target is not None:
            return getattr(importlib.import_module(target), name)
        if module == "tqdne" or module.startswith("tqdne."):
            raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which has no tqdne_amd counterpart")
        return super().find_class(module, name)


_REVERSE_MAP = {v: k for k, v in _MODULE_MAP.items()}


class _RemapPickler(pickle._Pickler):  # the pure-Python pickler: its save_global can be redirected
    def save_global(self, obj, name=None):
        mod = getattr(obj, "__module__", None)
        if isinstance(obj, type) and mod in _REVERSE_MAP:
            module_name, qual = _REVERSE_MAP[mod], obj.__qualname__
            if self.proto >= 4:
                self.save(module_name)
                self.save(qual)
                self.write(pickle.STACK_GLOBAL)
            else:
                self.write(pickle.GLOBAL + module_name.encode() + b"\n" + qual.encode() + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)


class _RemapPickle:
    """``pickle_module`` for ``torch.load`` / ``torch.save``: the standard wire format, with classes of this package written
    under (and read from) the reference's module paths."""

    __name__ = "tqdne_amd_checkpoint_pickle"
    Unpickler = _RemapUnpickler
    Pickler = _RemapPickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    DEFAULT_PROTOCOL = pickle.DEFAULT_PROTOCOL

    @staticmethod
    def load(f, **kw):
        return _RemapUnpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        return _RemapUnpickler(io.BytesIO(b), **kw).load()

    @staticmethod
    def dump(obj, f, protocol=None, **kw):
        _RemapPickler(f, protocol).dump(obj)

    @staticmethod
    def dumps(obj, protocol=None, **kw):
        bio = io.BytesIO()
        _RemapPickler(bio, protocol).dump(obj)
        return bio.getvalue()


def load_checkpoint(path, map_location="cpu") -> Dict[str, Any]:
    """The raw checkpoint dict of a reference (or tqdne_amd) ``.ckpt`` file."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_RemapPickle)


def save_checkpoint(module: torch.nn.Module, path, *, ema_state: Optional[Dict[str, torch.Tensor]] = None,
                    optimizer: Optional[torch.optim.Optimizer] = None, lr_scheduler=None, epoch: int = 0,
                    global_step: int = 0) -> None:
    """Write ``module`` in the layout of a Lightning checkpoint of the reference class of the same name."""
    ckpt: Dict[str, Any] = OrderedDict()
    ckpt["epoch"] = int(epoch)
    ckpt["global_step"] = int(global_step)
    ckpt["pytorch-lightning_version"] = "2.5.1"  # the version the reference pins (uv.lock)
    ckpt["state_dict"] = OrderedDict((k, v.detach().cpu()) for k, v in module.state_dict().items())
    ckpt["optimizer_states"] = [optimizer.state_dict()] if optimizer is not None else []
    ckpt["lr_schedulers"] = [lr_scheduler.state_dict()] if lr_scheduler is not None else []
    ckpt["hparams_name"] = "kwargs"
    ckpt["hyper_parameters"] = dict(getattr(module, "hparams", {}) or {})
    if ema_state is not None:
        ckpt["ema_state"] = OrderedDict((k, v.detach().cpu()) for k, v in ema_state.items())
    torch.save(ckpt, path, pickle_module=_RemapPickle)


def apply_ema(module: torch.nn.Module, ckpt: Dict[str, Any]) -> None:
    """Load the EMA weights the way the reference does before validation / prediction (tqdne/ema.py:30-32)."""
    if "ema_state" not in ckpt:
        raise KeyError("checkpoint holds no 'ema_state' (it was trained without the EMA callback)")
    module.load_state_dict(ckpt["ema_state"], strict=False)
