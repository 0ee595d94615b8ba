#!/usr/bin/env python3
"""Run ONE data-gradient shape repeatedly, with its chain epilogue (x SiLU'(a x + s), GroupNorm-backward sums), for rocprofv3 --pmc.
usage: bench_dgrad_one.py C_dy C_dx K T [B] [reps] [scheme: mx6 | bf16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops, _lib
Cdy, Cdx, K, T = map(int, sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
wfmt = _lib.TQ_WFMT_BF16X3 if (len(sys.argv) > 7 and sys.argv[7] == "bf16x3") else _lib.TQ_WFMT_F16_MX6
dev = torch.device("cuda:0")
dy = torch.randn(B, T, Cdy, device=dev) * 1e-5
w = torch.randn(Cdy, Cdx, K, device=dev) / (K * Cdx) ** 0.5
x = torch.randn(B, T, Cdx, device=dev)
gs, gh = torch.rand(B, Cdx, device=dev) + 0.5, torch.randn(B, Cdx, device=dev)
amax = ops.amax_bits(dy) if wfmt == _lib.TQ_WFMT_F16_MX6 else None
for _ in range(reps):
    ops.conv1d_bwd_data(dy, w, x0=x, gscale=gs, gshift=gh, silu=True, stats=True, wfmt=wfmt, dy_amax=amax)
torch.cuda.synchronize()
