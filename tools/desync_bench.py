#!/usr/bin/env python3
"""Experiment: does running co-resident workgroups OUT of phase pay?  One fused conv over B=64 as a single launch (all
workgroups in lockstep) against the same work as 4 launches of B=16 alternating on two streams, the second stream started half
a kernel later, so that each CU hosts workgroups of two launches in different phases.  usage: desync_bench.py C0 Cout K T"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tqdne_amd import ops, _lib

C0, Co, K, T = map(int, sys.argv[1:5])
B = 64
dev = torch.device("cuda:0")
lib = _lib.load()
x = torch.randn(B, T, C0, device=dev)
w = torch.randn(Co, C0, K, device=dev) / (K * C0) ** 0.5
b = torch.randn(Co, device=dev)
gs, gh = torch.rand(B, C0, device=dev) + 0.5, torch.randn(B, C0, device=dev)
y = torch.empty(B, T, Co, device=dev)
st = torch.empty(B, (T + 127) // 128, Co, 2, device=dev)
wp = ops.pack_conv_weight(w, 0)
p = lambda t: t.data_ptr()


def desc(nb):
    d = _lib.TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = nb, T, T, C0, 0, Co
    d.ktaps, d.stride, d.pad, d.upsample, d.flags = K, 1, K // 2, 0, 3 | 16
    return d


def launch(d, b0, nb, stream):
    rc = lib.tq_conv1d_fwd(C.byref(d), p(x[b0:]), None, p(gs[b0:]), p(gh[b0:]), p(wp), p(b), None, None, p(y[b0:]), p(st[b0:]),
                           stream.cuda_stream)
    assert rc == 0, rc


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


s0 = torch.cuda.current_stream()
d64 = desc(64)
t_one = timeit(lambda: launch(d64, 0, 64, s0))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for parts in (2, 4, 8):
    nb = B // parts
    dq = desc(nb)

    def split():
        # parts launches alternate between two streams; back-to-back launches on one stream keep that stream's slot busy
        for i in range(parts):
            launch(dq, i * nb, nb, sa if i % 2 == 0 else sb)
    t_split = timeit(split)
    print(f"C{C0}->{Co} k{K} T{T}: one launch {t_one:7.1f} us | {parts} launches of B={nb} on two streams {t_split:7.1f} us  ({t_one / t_split:.2f}x)")
