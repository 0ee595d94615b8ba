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


# classes of the mapped modules a checkpoint may instantiate (hyper_parameters hold a pickled tqdne.edm.EDM; the reference
# allow-lists exactly that class with add_safe_globals, experiments/generate.py:117-120)
_MAPPED_CLASSES = {"EDM", "LightningEDM", "LithningConsistencyModel", "LightningAutoencoder", "UNetModel", "Encoder", "Decoder"}

# everything else a Lightning checkpoint of the reference is made of: tensor / storage rebuild helpers, dtypes, containers,
# numpy scalars.  A global outside this list is refused: unpickling never runs code a downloaded .ckpt chooses.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._utils", "_rebuild_qtensor"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    # NOT torch.storage._load_from_bytes: it is torch.load(BytesIO(b), weights_only=False), i.e. an unrestricted unpickler
    # reachable through a bytes argument (torch's own weights_only unpickler excludes it for the same reason).  Zip-format
    # Lightning checkpoints never need it.
    ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
    ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"),
}
_SAFE_TORCH_NAMES = {n for n in dir(torch) if n.endswith("Storage")} | {
    n for n in dir(torch) if isinstance(getattr(torch, n, None), torch.dtype)}
_SAFE_BUILTINS = {"set", "frozenset", "tuple", "list", "dict", "int", "float", "bool", "str", "bytes", "complex", "slice", "range",
                  "bytearray"}


class _RemapUnpickler(pickle.Unpickler):
    """Restricted unpickler (the allow-list idea of ``torch.load(weights_only=True)`` + ``add_safe_globals``, which is how the
    reference loads, plus the tqdne.* -> tqdne_amd.* remapping).  ``extra`` = additional (module, name) pairs to accept."""

    extra = frozenset()

    def find_class(self, module: str, name: str):
        target = _MODULE_MAP.get(module)
        if target is not None:
            if name not in _MAPPED_CLASSES:
                raise pickle.UnpicklingError(f"checkpoint references {module}.{name}: not a class a tqdne checkpoint may hold")
            return getattr(importlib.import_module(target), name)
        if module == "tqdne" or module.startswith("tqdne."):
            raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which has no tqdne_amd counterpart")
        ok = ((module, name) in _SAFE_GLOBALS or (module, name) in self.extra
              or (module == "torch" and name in _SAFE_TORCH_NAMES) or (module == "builtins" and name in _SAFE_BUILTINS))
        if not ok:
            raise pickle.UnpicklingError(
                f"refusing to unpickle global {module}.{name} from a checkpoint (not on the allow-list; pass "
                f"extra_safe_globals=[({module!r}, {name!r})] to load_checkpoint if the file is trusted)")
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


def load_checkpoint(path, map_location="cpu", extra_safe_globals=()) -> Dict[str, Any]:
    """The raw checkpoint dict of a reference (or tqdne_amd) ``.ckpt`` file.  Unpickling is restricted to an allow-list of
    tensor / container globals and the remapped tqdne classes (see ``_RemapUnpickler``); ``extra_safe_globals``: further
    (module, name) pairs to accept for a trusted file."""
    if not extra_safe_globals:
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_RemapPickle)

    class _U(_RemapUnpickler):
        extra = frozenset(tuple(x) for x in extra_safe_globals)

    class _P(_RemapPickle):
        Unpickler = _U

        @staticmethod
        def load(f, **kw):
            return _U(f, **kw).load()

        @staticmethod
        def loads(b, **kw):
            return _U(io.BytesIO(b), **kw).load()

    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_P)


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
