#!/usr/bin/env python3
"""Time the attention kernels on the paper shape (B=64, T=512, 4 heads x 64) and report their error against an fp64 torch
evaluation of the reference's QKVAttention (tqdne/blocks.py:156-190).  usage: bench_attention.py [B] [T] [H] [D] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import ops

B, T, H, D, reps = (list(map(int, sys.argv[1:])) + [64, 512, 4, 64, 50][len(sys.argv) - 1:])[:5]
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * H * D, device=dev) * 1.5
dout = torch.randn(B, T, H * D, device=dev)


def ref64(qkv):
    x = qkv.double().view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)  # (3, B, H, T, D)
    q, k, v = x[0], x[1], x[2]
    s = torch.einsum("bhtd,bhsd->bhts", q, k) / D ** 0.5
    return torch.einsum("bhts,bhsd->bhtd", s.softmax(-1), v).permute(0, 2, 1, 3).reshape(B, T, H * D)


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out, lse = ops.attention(qkv, H, return_lse=True)
nb = min(B, 4)
q64 = qkv[:nb].double().requires_grad_(True)
B0, B = B, nb
r = ref64(q64)
r.backward(dout[:nb].double())
B = B0
err = ((out[:nb].double() - r).abs().max() / r.abs().max()).item()
dq = ops.attention_bwd(qkv, out, dout, lse, H)
gerr = ((dq[:nb].double() - q64.grad).abs().max() / q64.grad.abs().max()).item()
flops = 4.0 * T * T * H * D * B
t_f = timed(lambda: ops.attention(qkv, H, return_lse=True))
t_b = timed(lambda: ops.attention_bwd(qkv, out, dout, lse, H))
t_b1 = timed(lambda: ops.attention_bwd(qkv, out, dout, lse, H, workspace=False))
dq1 = ops.attention_bwd(qkv, out, dout, lse, H, workspace=False)
gerr1 = ((dq1[:nb].double() - q64.grad).abs().max() / q64.grad.abs().max()).item()
print(f"attention B={B} T={T} H={H} D={D}: fwd(+prep) {t_f:.1f} us ({flops / t_f * 1e-6:.0f} TFLOP/s)  bwd {t_b:.1f} us "
      f"({2.5 * flops / t_b * 1e-6:.0f} TFLOP/s; first generation {t_b1:.1f} us)  max err fwd {err:.2e} bwd {gerr:.2e} (first generation {gerr1:.2e})")
