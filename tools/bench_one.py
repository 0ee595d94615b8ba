#!/usr/bin/env python3
"""Run ONE fused-conv shape repeatedly (for rocprofv3 --pmc).  usage: bench_one.py C0 C1 Cout K T [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tqdne_amd import ops, _lib
C0, C1, Co, K, T = map(int, sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 64
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
lib = _lib.load()
x0 = torch.randn(B, T, C0, device=dev)
x1 = torch.randn(B, T, C1, device=dev) if C1 else None
w = torch.randn(Co, C0 + C1, K, device=dev) / (K * (C0 + C1)) ** 0.5
b = torch.randn(Co, device=dev)
gs = torch.rand(B, C0 + C1, device=dev) + 0.5
gh = torch.randn(B, C0 + C1, device=dev)
y = torch.empty(B, T, Co, device=dev)
st = torch.empty(B, (T + 31) // 32, Co, 2, device=dev)   # (room for the 32-position slots of a small-tile launch)
wfmt = _lib.forward_wfmt(Co, [C0, C1])
wp = ops.pack_conv_weight(w, _lib.PACK_MODE[wfmt])
p = lambda t: None if t is None else t.data_ptr()
d = _lib.TqConvDesc()
d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
d.ktaps, d.stride, d.pad, d.upsample, d.flags = K, 1, K // 2, 0, 3 | 16
d.wfmt = wfmt
if os.environ.get("TQ_TTILE"):   # (TQ_TTILE=32: the small tile of launch-bound plans; TQ_FOLD=1: with its own GroupNorm fold)
    d.t_tile = int(os.environ["TQ_TTILE"])
fold = None
if os.environ.get("TQ_FOLD") == "1":
    ns = (T + 127) // 128
    st0 = torch.randn(B, ns, C0, 2, device=dev).abs() + 1.0
    st0[..., 1] = st0[..., 0] ** 2 / 128 + 50.0
    gamma, beta = torch.rand(C0 + C1, device=dev) + 0.5, torch.randn(C0 + C1, device=dev)
    mr = torch.empty(B, 32, 2, device=dev)
    fold = _lib.TqGnFold()
    fold.stats0, fold.stats1, fold.slot0, fold.slot1 = p(st0), None, 128, 0
    fold.gamma, fold.beta, fold.mean_rstd = p(gamma), p(beta), p(mr)
    assert C1 == 0
    d.gn_fold = C.pointer(fold)
for _ in range(reps):
    assert lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(b), None, None, p(y), p(st), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
