#!/usr/bin/env python3
"""Timing of one conv launch with the library given in TQDNE_HIP_LIB (developer tool for conv1d_w4.hip ablation builds).
usage: python tools/experiments/w4_time.py C0 C1 Cout T [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tqdne_amd import _lib, ops
C0, C1, Co, T = map(int, sys.argv[1:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
lib = _lib.load(); dev = torch.device("cuda:0")
x0 = torch.randn(B, T, C0, device=dev); x1 = torch.randn(B, T, C1, device=dev) if C1 else None
w = torch.randn(Co, C0 + C1, 5, device=dev) / (5 * (C0 + C1)) ** 0.5
bias = torch.randn(Co, device=dev); gs = torch.rand(B, C0 + C1, device=dev) + 0.5; gh = torch.randn(B, C0 + C1, device=dev)
y = torch.empty(B, T, Co, device=dev); st = torch.zeros(B, (T + 127) // 128, Co, 2, device=dev)
wp = ops.pack_conv_weight(w, _lib.PACK_MODE[_lib.TQ_WFMT_F16_MX6])
d = _lib.TqConvDesc(); d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
d.ktaps, d.stride, d.pad, d.upsample, d.flags, d.wfmt = 5, 1, 2, 0, 1 | 2 | 16, _lib.TQ_WFMT_F16_MX6
p = lambda t: None if t is None else t.data_ptr(); s = torch.cuda.current_stream().cuda_stream
run = lambda: lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(bias), None, None, p(y), p(st), s)
for _ in range(5): assert run() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
for _ in range(7):
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
print(f"{os.path.basename(os.environ.get('TQDNE_HIP_LIB', 'default'))} W4={os.environ.get('TQDNE_CONV_W4', '1')} {C0}+{C1}->{Co} T{T} B{B}: {sorted(ts)[3]:.1f} us")
