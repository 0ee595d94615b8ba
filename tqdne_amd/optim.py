"""Fused Adam (+ EMA) for the training step: one HIP launch over the whole model (tq_adam_ema_step).

Semantics are ``torch.optim.Adam(params, lr)`` as configured by the reference (tqdne/edm.py:240-251: default betas / eps, no
weight decay) and, when ``ema_decay`` is given, the reference's EMA callback (tqdne/ema.py:24-28: ``ema.lerp_(p, 1 - decay)``
after every optimizer step).  The class is a ``torch.optim.Optimizer``: LR schedulers drive ``param_groups[0]["lr"]`` and
``state_dict()`` / ``load_state_dict()`` speak torch Adam's format, so the ``optimizer_states`` of a reference checkpoint load.
Moments (and the EMA copy) live in flat fp32 buffers; ``state[p]["exp_avg"]`` etc. are views of them.
"""

from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import TQ_ADAM_CHUNK, TqAdamChunk, check


class FusedAdamEMA(torch.optim.Optimizer):
    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, ema_decay: Optional[float] = None, weight_decay: float = 0.0):
        """``weight_decay`` > 0 gives torch.optim.AdamW (decoupled decay), the autoencoder's optimizer (autoencoder.py:93-95)."""
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        self._names = [n for n, _ in named]
        super().__init__([p for _, p in named], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.ema_decay = ema_decay
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdamEMA updates parameters on the GPU; move the module to cuda first")
        self._lib = _lib.load()  # raises when the HIP library is missing: there is no fallback update
        offs, total = [], 0
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FusedAdamEMA handles contiguous fp32 parameters on one device")
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64  # every tensor starts 256-byte aligned
        self._offs = offs
        self._m = torch.zeros(total, device=dev)
        self._v = torch.zeros(total, device=dev)
        self._ema = None
        if ema_decay is not None:
            self._ema = torch.zeros(total, device=dev)
            for p, o in zip(ps, offs):
                self._ema[o:o + p.numel()].copy_(p.detach().reshape(-1))
        self._step = 0
        self._step_t = torch.tensor(0.0)  # one shared CPU scalar: state[p]["step"] of every parameter (torch Adam's format)
        for p, o in zip(ps, offs):
            n = p.numel()
            self.state[p] = dict(step=self._step_t, exp_avg=self._m[o:o + n].view_as(p),
                                 exp_avg_sq=self._v[o:o + n].view_as(p))
        self._table = None
        self._table_key = None

    # ------------------------------------------------------------------ EMA
    def ema_state(self) -> "OrderedDict[str, torch.Tensor]":
        """name -> EMA tensor (views), the dict the reference's EMA callback stores under checkpoint['ema_state']."""
        if self._ema is None:
            raise RuntimeError("constructed without ema_decay")
        ps = self.param_groups[0]["params"]
        return OrderedDict((n, self._ema[o:o + p.numel()].view_as(p)) for n, p, o in zip(self._names, ps, self._offs))

    def load_ema_state(self, ema_state):
        for n, t in self.ema_state().items():
            t.copy_(ema_state[n])

    # ------------------------------------------------------------------ chunk table
    def _build_table(self):
        ps = self.param_groups[0]["params"]
        rows = []
        for p, o in zip(ps, self._offs):
            if p.grad is None:
                continue  # like torch: parameters without a gradient are skipped
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("FusedAdamEMA needs contiguous fp32 gradients")
            n = p.numel()
            for s in range(0, n, TQ_ADAM_CHUNK):
                c = min(TQ_ADAM_CHUNK, n - s)
                rows.append((p.data_ptr() + 4 * s, g.data_ptr() + 4 * s, self._m.data_ptr() + 4 * (o + s),
                             self._v.data_ptr() + 4 * (o + s), 0 if self._ema is None else self._ema.data_ptr() + 4 * (o + s), c))
        arr = (TqAdamChunk * len(rows))()
        for i, r in enumerate(rows):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].ema, arr[i].n = r[0], r[1], r[2], r[3], r[4] or None, r[5]
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._table = host.to(self._m.device)
        self._n_chunks = len(rows)
        self._updated = [p for p in ps if p.grad is not None]

    def _key(self):
        ps = self.param_groups[0]["params"]
        return tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr()) for p in ps)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, skip_flag: Optional[torch.Tensor] = None):
        """``skip_flag``: device int32 tensor; when it is non-zero at execution time the launch changes nothing (the range guard
        of the fp16-range forward scheme, see tq_adam_ema_step_guarded) -- decided on the device, no host synchronisation."""
        loss = closure() if closure is not None else None
        key = self._key()
        if key != self._table_key:
            self._build_table()
            self._table_key = key
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self._step += 1
        t = self._step
        step_size = g["lr"] / (1.0 - b1 ** t)
        ibc2 = 1.0 / math.sqrt(1.0 - b2 ** t)
        ema_w = 0.0 if self.ema_decay is None else 1.0 - self.ema_decay
        stream = torch.cuda.current_stream(self._m.device).cuda_stream
        check(self._lib.tq_adam_ema_step_guarded(self._table.data_ptr(), self._n_chunks, step_size, b1, b2, g["eps"], ibc2, ema_w,
                                                 grad_scale, 1.0 - g["lr"] * g["weight_decay"],
                                                 None if skip_flag is None else skip_flag.data_ptr(), stream), "adam")
        # the kernel wrote the parameters through raw pointers: bump their autograd version counters so that everything keyed
        # on ``p._version`` (the engines' packed MFMA weight fragments, engine.py repack / repack_transposed) sees the update
        torch._C._increment_version(self._updated)
        self._step_t.fill_(float(t))
        return loss

    # ------------------------------------------------------------------ (de)serialisation in torch Adam's format
    def load_state_dict(self, state_dict):
        ps = self.param_groups[0]["params"]
        st = state_dict["state"]
        steps = set()
        for i, p in enumerate(ps):
            if i in st:
                self.state[p]["exp_avg"].copy_(st[i]["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st[i]["exp_avg_sq"])
                steps.add(int(float(st[i]["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused update keeps one")
        self._step = steps.pop() if steps else 0
        self._step_t.fill_(float(self._step))
        for k, v in state_dict["param_groups"][0].items():
            if k in ("lr", "betas", "eps", "initial_lr", "weight_decay"):
                self.param_groups[0][k] = v
