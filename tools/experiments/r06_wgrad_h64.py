#!/usr/bin/env python3
"""Round 6: weight gradients of the convs with 64 output channels (paper UNet, T = 4096 level), each launch alone, 20 repetitions between
two HIP events.  Run once per setting of TQDNE_WGRAD_H64 / TQDNE_WGRAD_SLOTS (read once per process).  usage: r06_wgrad_h64.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tqdne_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
# (C0, C1, Cout, K, T): input_blocks.1-2 (both convs), output_blocks.9 conv1, output_blocks.10-11 conv1, output_blocks.9-11 conv2; one
# 128-channel shape as the control (never takes the new form)
LAYERS = [(64, 0, 64, 5, 4096), (128, 64, 64, 5, 4096), (64, 64, 64, 5, 4096), (128, 0, 128, 5, 2048)]
tot = 0.0
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
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    fl = 2.0 * (C0 + C1) * Co * K * T * B
    tot += us if Co == 64 else 0.0
    print(f"wgrad {C0}+{C1} -> {Co} k{K} T{T}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic  (incl. the allocation of the slab and the reduce launch)")
print(f"sum of the 64-channel shapes: {tot:.1f} us")
