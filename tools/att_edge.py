#!/usr/bin/env python3
"""Attention forward / backward (both generations) against fp64 autograd on edge lengths: T in {1, 5, 17, 63, 64, 65, 129, 191}, D in
{32, 64}.  (developer check, GPU box; round 3: worst 2.7e-5)"""
import sys, math, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda:0")
worst = 0
for D in (32, 64):
    for T in (1, 5, 17, 63, 64, 65, 129, 191):
        B, H = 2, 2
        qkv = (torch.randn(B, T, 3 * H * D, device=dev) * 1.3)
        dout = torch.randn(B, T, H * D, device=dev)
        x = qkv.double().view(B, T, 3, H, D).permute(2, 0, 3, 1, 4).detach().requires_grad_(True)
        s = torch.einsum("bhtd,bhsd->bhts", x[0], x[1]) / D ** 0.5
        r = torch.einsum("bhts,bhsd->bhtd", s.softmax(-1), x[2]).permute(0, 2, 1, 3).reshape(B, T, H * D)
        r.backward(dout.double())
        g = x.grad.permute(1, 3, 0, 2, 4).reshape(B, T, 3 * H * D)
        out, lse = ops.attention(qkv, H, return_lse=True)
        e1 = float((out.double() - r).abs().max() / r.abs().max())
        dq = ops.attention_bwd(qkv, out, dout, lse, H)
        e2 = float((dq.double() - g).abs().max() / g.abs().max())
        dq1 = ops.attention_bwd(qkv, out, dout, lse, H, workspace=False)
        e3 = float((dq1.double() - g).abs().max() / g.abs().max())
        worst = max(worst, e1, e2, e3)
        print(D, T, f"{e1:.1e} {e2:.1e} {e3:.1e}", bool(torch.isfinite(dq).all()))
print("worst", worst)
