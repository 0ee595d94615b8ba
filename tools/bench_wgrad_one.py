#!/usr/bin/env python3
"""Run ONE weight-gradient shape repeatedly (for rocprofv3 --pmc).  usage: bench_wgrad_one.py C0 C1 Cout K T [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops
C0, C1, Co, K, T = map(int, sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 64
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
x0 = torch.randn(B, T, C0, device=dev)
x1 = torch.randn(B, T, C1, device=dev) if C1 else None
dy = torch.randn(B, T, Co, device=dev)
gs, gh = torch.rand(B, C0 + C1, device=dev) + 0.5, torch.randn(B, C0 + C1, device=dev)
for _ in range(reps):
    ops.conv1d_bwd_weight(dy, x0, (Co, C0 + C1, K), x1=x1, gscale=gs, gshift=gh, silu=True)
torch.cuda.synchronize()
