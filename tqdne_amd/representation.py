"""Signal representations either side of the diffusion path (mirror of tqdne/representation.py).

``MovingAverageEnvelope`` is the representation of the reference's 1-D experiments (experiments/config.py:62-67): it turns a
3-channel waveform into the 6-channel signal the EDM is trained on and back.  The reference computes it with
``np.apply_along_axis(np.convolve)`` on the host (a loader bottleneck and a post-sampling step); here both directions are one
HBM-bound HIP launch (tq_envelope_fwd / tq_envelope_inv) on tensors that already live on the GPU.

Interface: same class and method names and arguments.  Inputs may be torch tensors on a GPU (returned as GPU tensors) or numpy
arrays / CPU tensors (uploaded, computed on the GPU, returned as numpy like the reference's NumpyArgMixin does).  Results are
fp32 (computed in float64 inside the kernel, as numpy does; the reference returns float64 arrays that its Dataset casts to
fp32).  There is no host fallback: without a GPU and the HIP library the calls raise.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check


def _to_gpu(x):
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.contiguous().float(), "cuda"
    if not torch.cuda.is_available():
        raise RuntimeError("tqdne_amd.representation runs on the GPU (no CPU fallback): no GPU is visible")
    kind = "torch" if isinstance(x, torch.Tensor) else "numpy"
    t = torch.as_tensor(np.asarray(x) if kind == "numpy" else x, dtype=torch.float32)
    return t.contiguous().to("cuda"), kind


def _back(y, kind):
    if kind == "cuda":
        return y
    return y.cpu() if kind == "torch" else y.cpu().numpy()


class Representation:
    def get_representation(self, waveform):
        raise NotImplementedError

    def invert_representation(self, representation):
        raise NotImplementedError


class Identity(Representation):  # representation.py:21-26
    def get_representation(self, waveform):
        return waveform

    def invert_representation(self, representation):
        return representation


class Normalization(Representation):  # representation.py:29-38
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def get_representation(self, waveform):
        return (waveform - self.mean) / self.std

    def invert_representation(self, representation):
        return representation * self.std + self.mean


class MovingAverageEnvelope(Representation):  # representation.py:41-60
    def __init__(self, window_size=128, log_eps=1e-6, eps=1e-6):
        self.window_size, self.log_eps, self.eps = window_size, log_eps, eps

    def get_representation(self, waveform):
        x, kind = _to_gpu(waveform)
        if x.dim() < 2:
            raise ValueError("expected (..., channels, time)")
        C_, T = x.shape[-2], x.shape[-1]
        if T < self.window_size:
            raise ValueError(f"signal length {T} < window {self.window_size} (np.convolve(mode='same') would change the length)")
        N = x.numel() // (C_ * T)
        out = torch.empty(x.shape[:-2] + (2 * C_, T), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        check(lib.tq_envelope_fwd(x.data_ptr(), out.data_ptr(), N, C_, T, self.window_size, float(self.log_eps), float(self.eps),
                                  torch.cuda.current_stream(x.device).cuda_stream), "envelope_fwd")
        return _back(out, kind)

    def invert_representation(self, representation):
        r, kind = _to_gpu(representation)
        C2, T = r.shape[-2], r.shape[-1]
        if C2 % 2:
            raise ValueError("representation must have an even number of channels (scaled waveform | log envelope)")
        C_ = C2 // 2
        N = r.numel() // (C2 * T)
        out = torch.empty(r.shape[:-2] + (C_, T), device=r.device, dtype=torch.float32)
        lib = _lib.load()
        check(lib.tq_envelope_inv(r.data_ptr(), out.data_ptr(), N, C_, T, float(self.log_eps), float(self.eps),
                                  torch.cuda.current_stream(r.device).cuda_stream), "envelope_inv")
        return _back(out, kind)
