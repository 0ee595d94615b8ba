#!/usr/bin/env python3
"""Micro-benchmark of the fused conv launches of the paper UNet (B=64) -- developer tool, run on the GPU box.
usage: [TQDNE_HIP_LIB=...] python tools/bench_conv.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops, _lib
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
# (C0, C1, Cout, K, T, prologue: 0 none, 1 GN + SiLU, 2 GN only)
LAYERS = [(64, 0, 64, 5, 4096, 1), (128, 0, 128, 5, 2048, 1), (256, 0, 256, 5, 1024, 1), (256, 0, 256, 5, 512, 1),
          (256, 256, 256, 5, 1024, 1), (256, 128, 256, 5, 1024, 1), (256, 256, 256, 1, 1024, 0), (256, 0, 768, 1, 512, 1),
          (128, 64, 128, 5, 2048, 1), (128, 64, 64, 5, 4096, 1),
          (256, 0, 768, 1, 512, 2), (256, 0, 256, 1, 512, 0)]  # the attention block's qkv / proj_out
lib = _lib.load()
for (C0, C1, Co, K, T, gn) in LAYERS:
    x0 = torch.randn(B, T, C0, device=dev)
    x1 = torch.randn(B, T, C1, device=dev) if C1 else None
    w = torch.randn(Co, C0 + C1, K, device=dev) / (K * (C0 + C1)) ** 0.5
    b = torch.randn(Co, device=dev)
    gs = torch.rand(B, C0 + C1, device=dev) + 0.5 if gn else None
    gh = torch.randn(B, C0 + C1, device=dev) if gn else None
    y = torch.empty(B, T, Co, device=dev)
    st = torch.empty(B, (T + 127) // 128, Co, 2, device=dev)
    d_wfmt = _lib.forward_wfmt(Co, [C0, C1])
    wp = ops.pack_conv_weight(w, _lib.PACK_MODE[d_wfmt])
    d = _lib.TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
    d.ktaps, d.stride, d.pad, d.upsample = K, 1, K // 2, 0
    d.flags = {0: 0, 1: 3, 2: 1}[gn] | 16
    d.wfmt = d_wfmt
    stream = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    def run():
        rc = lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(b), None, None, p(y), p(st), stream)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 20
    e0.record()
    for _ in range(N):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    fl = 2.0 * (C0 + C1) * Co * K * T * B
    print(f"C {C0}+{C1}->{Co} k{K} T{T}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic ({3 * fl / us / 1e6:7.1f} executed)  "
          f"in+out {(4.0 * B * T * (C0 + C1 + Co)) / us / 1e3:7.0f} GB/s")
