#!/usr/bin/env python3
"""Run the attention forward + backward of one paper-UNet attention block repeatedly (for rocprofv3).  usage: [B] [T] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, D = 4, 64
dev = torch.device("cuda:0")
qkv = torch.randn(B, T, 3 * H * D, device=dev)
dout = torch.randn(B, T, H * D, device=dev)
for _ in range(reps):
    o, lse = ops.attention(qkv, H, return_lse=True)
    ops.attention_bwd(qkv, o, dout, lse, H)
torch.cuda.synchronize()
