#!/usr/bin/env python3
"""Stem / head kernels of the paper UNet at B = 64, T = 4096 (3 -> 64 and 64 -> 3 channels, k = 5): time per launch, median of 7 x 20
launches.  usage: [TQDNE_HIP_LIB=...] python tools/bench_ends.py [B]   (developer tool, GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
T = 4096
x = torch.randn(B, 3, T, device=dev)
w = torch.randn(64, 3, 5, device=dev) / 4
b = torch.randn(64, device=dev)
sc = torch.rand(B, device=dev) + 0.5
h = torch.randn(B, T, 64, device=dev)
hw = torch.randn(3, 64, 5, device=dev) / 18
hb = torch.randn(3, device=dev)
gs, gh = torch.rand(B, 64, device=dev) + 0.5, torch.randn(B, 64, device=dev)
co, cs = torch.rand(B, device=dev), torch.rand(B, device=dev)


def timed(fn):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[3]


print(f"stem 3->64 k5 B={B} T={T}: {timed(lambda: ops.stem_conv(x, w, b, in_scale=sc)):.1f} us")
print(f"head 64->3 k5 B={B} T={T}: {timed(lambda: ops.head_conv(h, hw, hb, gscale=gs, gshift=gh, c_out=co, c_skip=cs, skip_src=x)):.1f} us")
