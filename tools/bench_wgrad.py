#!/usr/bin/env python3
"""Time tq_conv1d_bwd_weight on the paper UNet's layer shapes (B = 64) -- developer tool, run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
# (C0, C1, Cout, K, T)
LAYERS = [(64, 0, 64, 5, 4096), (128, 0, 128, 5, 2048), (256, 0, 256, 5, 1024), (256, 0, 256, 5, 512), (256, 256, 256, 5, 1024),
          (256, 128, 256, 5, 1024), (128, 64, 64, 5, 4096), (256, 0, 768, 1, 512), (256, 256, 256, 1, 1024), (64, 0, 128, 5, 2048)]
for (C0, C1, Co, K, T) in LAYERS:
    x0 = torch.randn(B, T, C0, device=dev)
    x1 = torch.randn(B, T, C1, device=dev) if C1 else None
    dy = torch.randn(B, T, Co, device=dev)
    gs, gh = torch.rand(B, C0 + C1, device=dev) + 0.5, torch.randn(B, C0 + C1, device=dev)
    run = lambda: ops.conv1d_bwd_weight(dy, x0, (Co, C0 + C1, K), x1=x1, gscale=gs, gshift=gh, silu=True)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * (C0 + C1) * Co * K * T * B
    print(f"wgrad C {C0}+{C1}->{Co} k{K} T{T}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic")
