#!/usr/bin/env python3
"""Kernel-only time of the attention forward on pre-split K / V planes (the inference path: `tq_attention_fwd_presplit`), B = 64,
T = 512, 4 heads x 64, plus its error against fp64.  usage: att_time.py [B] [T] [H] [D] [reps]   (TQDNE_HIP_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tqdne_amd import _lib, ops
from tqdne_amd.ops import _p, _stream, check

B, T, H, D, reps = (list(map(int, sys.argv[1:])) + [64, 512, 4, 64, 100][len(sys.argv) - 1:])[:5]
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * H * D, device=dev) * 1.5
lib = _lib.load()
ws = torch.empty(lib.tq_attention_workspace_bytes(B, T, H, D), dtype=torch.uint8, device=dev)
out = torch.empty(B, T, H * D, device=dev)
check(lib.tq_attention_fwd(_p(qkv), _p(out), None, _p(ws), B, T, H, D, _stream(dev)), "attention")  # fills the planes
st = _stream(dev)


def run():
    check(lib.tq_attention_fwd_presplit(_p(qkv), _p(ws), _p(out), B, T, H, D, 0, st), "attention presplit")


for _ in range(10):
    run()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / reps * 1e3)
ts.sort()
nb = min(B, 2)
x = qkv[:nb].double().view(nb, T, 3, H, D).permute(2, 0, 3, 1, 4)
s = torch.einsum("bhtd,bhsd->bhts", x[0], x[1]) / D ** 0.5
r = torch.einsum("bhts,bhsd->bhtd", s.softmax(-1), x[2]).permute(0, 2, 1, 3).reshape(nb, T, H * D)
err = ((out[:nb].double() - r).abs().max() / r.abs().max()).item()
print(f"attention core B={B} T={T} H={H} D={D}: median {ts[3]:.1f} us (min {ts[0]:.1f})  max err {err:.2e}")
