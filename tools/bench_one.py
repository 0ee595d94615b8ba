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
st = torch.empty(B, (T + 127) // 128, Co, 2, device=dev)
wfmt = _lib.forward_wfmt(Co, [C0, C1])
wp = ops.pack_conv_weight(w, _lib.PACK_MODE[wfmt])
d = _lib.TqConvDesc()
d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
d.ktaps, d.stride, d.pad, d.upsample, d.flags = K, 1, K // 2, 0, 3 | 16
d.wfmt = wfmt
p = lambda t: None if t is None else t.data_ptr()
for _ in range(reps):
    assert lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(b), None, None, p(y), p(st), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
