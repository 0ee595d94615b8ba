#!/usr/bin/env python3
"""Round-4 micro-benchmarks of the training-backward kernels, B = 64, HIP events, median of 9 after 3 warm-ups (GPU box):
  * data gradient bf16x3 vs fp16 + MX-fp6 on a scaled dy (tq_conv1d_bwd_data, TqConvBwdDesc.wfmt), per layer shape
  * weight gradient + slab reduce (tq_conv1d_bwd_weight)
  * tq_gn_bwd_apply + tq_colsum  vs  tq_gn_bwd_apply_colsum
usage: tools/bwd_micro.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tqdne_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
lib = _lib.load()
p = lambda t: None if t is None else t.data_ptr()
stream = lambda: torch.cuda.current_stream().cuda_stream


def med(fn, n=9, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(1e3 * a.elapsed_time(b))
    return sorted(ts)[n // 2]


ONLY = os.environ.get("MICRO_ONLY", "")
print(f"# B = {B}; us per launch")
print("## data gradient: C_dy -> C_dx, k, T: bf16x3 | f16+mx6 | ratio")
for (cdy, cdx, k, T) in [] if ONLY not in ("", "dgrad") else [(256, 256, 5, 1024), (256, 512, 5, 1024), (256, 256, 5, 512), (128, 128, 5, 2048), (256, 384, 5, 1024),
                         (128, 256, 5, 2048), (768, 256, 1, 512), (256, 256, 1, 512), (256, 128, 3, 1024)]:
    dy = torch.randn(B, T, cdy, device=dev) * 1e-5
    w = torch.randn(cdy, cdx, k, device=dev) / (k * cdx) ** 0.5
    dx = torch.empty(B, T, cdx, device=dev)
    am = ops.amax_bits(dy)
    res = []
    for wf in (0, 2):
        d = _lib.TqConvBwdDesc()
        d.B, d.T, d.C_dy, d.C_dx0, d.C_dx1, d.ktaps, d.flags, d.wfmt = B, T, cdy, cdx, 0, k, 0, wf
        d.dy_amax = am.data_ptr() if wf else None
        wp = ops.pack_conv_weight(w, _lib.PACK_MODE_T[wf])
        def run():
            rc = lib.tq_conv1d_bwd_data(C.byref(d), p(dy), p(wp), None, None, None, None, p(dx), None, None, stream())
            assert rc == 0, rc
        res.append(med(run))
    print(f"{cdy:4d} -> {cdx:4d} k{k} T{T:5d}: {res[0]:7.1f} | {res[1]:7.1f} | {res[1] / res[0]:.2f}")

print("## weight gradient (+ slab reduce): C_in -> C_out, k, T")
for (ci, co, k, T) in [] if ONLY not in ("", "wgrad") else [(256, 256, 5, 1024), (512, 256, 5, 1024), (256, 256, 5, 512), (128, 128, 5, 2048), (64, 64, 5, 4096), (256, 768, 1, 512),
                       (512, 256, 1, 1024)]:
    x = torch.randn(B, T, ci, device=dev)
    dy = torch.randn(B, T, co, device=dev)
    gs, gh = torch.rand(B, ci, device=dev) + 0.5, torch.randn(B, ci, device=dev)
    d = _lib.TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, ci, 0, co
    d.ktaps, d.stride, d.pad, d.upsample, d.flags = k, 1, k // 2, 0, 3
    ws = torch.empty(lib.tq_conv1d_bwd_weight_workspace(C.byref(d)), dtype=torch.uint8, device=dev)
    dw = torch.empty(co, ci, k, device=dev)
    def run():
        assert lib.tq_conv1d_bwd_weight(C.byref(d), p(dy), p(x), None, p(gs), p(gh), p(dw), p(ws), ws.numel(), stream()) == 0
    print(f"{ci:4d} -> {co:4d} k{k} T{T:5d}: {med(run):7.1f}")

print("## GroupNorm backward apply (+ column sums): C, T: apply | colsum | apply + colsum | fused")
for (Cc, T) in [] if ONLY not in ("", "gn") else [(64, 4096), (128, 2048), (256, 1024), (256, 512), (512, 1024)]:
    G, x, r = (torch.randn(B, T, Cc, device=dev) for _ in range(3))
    coefs = tuple(torch.randn(B, Cc, device=dev) for _ in range(3))
    dx = torch.empty_like(G)
    obc, oc = torch.zeros(B, Cc, device=dev), torch.zeros(Cc, device=dev)
    am = torch.zeros(_lib.TQ_AMAX_WORDS, dtype=torch.int32, device=dev)
    def apply():
        assert lib.tq_gn_bwd_apply(p(G), p(x), p(r), p(coefs[0]), p(coefs[1]), p(coefs[2]), p(dx), B, T, Cc, Cc, 0, 0, stream()) == 0
    def cs():
        assert lib.tq_colsum(p(dx), B, T, Cc, p(obc), Cc, p(oc), None, None, p(am), stream()) == 0
    def both():
        apply(); cs()
    def fused():
        assert lib.tq_gn_bwd_apply_colsum(p(G), p(x), p(r), p(coefs[0]), p(coefs[1]), p(coefs[2]), p(dx), B, T, Cc, Cc, 0, 0, p(obc), Cc, p(oc),
                                          None, p(am), stream()) == 0
    def cs_rows():     # per-sample sums only (no cross-sample same-address atomics)
        assert lib.tq_colsum(p(dx), B, T, Cc, p(obc), Cc, None, None, None, p(am), stream()) == 0
    def fused_rows():
        assert lib.tq_gn_bwd_apply_colsum(p(G), p(x), p(r), p(coefs[0]), p(coefs[1]), p(coefs[2]), p(dx), B, T, Cc, Cc, 0, 0, p(obc), Cc, None,
                                          None, p(am), stream()) == 0
    print(f"{Cc:4d} x {T:5d}: {med(apply):7.1f} | {med(cs):7.1f} | {med(both):7.1f} | {med(fused):7.1f}   per-sample sums only: colsum {med(cs_rows):7.1f}, fused {med(fused_rows):7.1f}")

print("## stem weight gradient | head backward (3 x 4096, 64 channels)   [TQDNE_STEM_HEAD_BWD=3: round 3's kernels]")
if ONLY in ("", "stemhead"):
    T = 4096
    dy = torch.randn(B, T, 64, device=dev); xin = torch.randn(B, 3, T, device=dev); sc = torch.rand(B, device=dev) + 0.5
    dw = torch.zeros(64, 3, 5, device=dev)
    wsb = torch.empty(lib.tq_stem_head_bwd_workspace(), dtype=torch.uint8, device=dev)
    use_ws = os.environ.get("MICRO_NO_WS") != "1"
    def stem():
        assert lib.tq_stem_conv_bwd_weight_ws(p(dy), p(xin), p(sc), p(dw), B, 3, T, 64, 5, p(wsb) if use_ws else None, wsb.numel() if use_ws else 0, stream()) == 0
    hh = torch.randn(B, T, 64, device=dev); ga, gs = torch.rand(B, 64, device=dev) + 0.5, torch.randn(B, 64, device=dev)
    wh = torch.randn(3, 64, 5, device=dev) / 10; dpred = torch.randn(B, 3, T, device=dev); co = torch.rand(B, device=dev) + 0.5
    Gh = torch.empty_like(hh); gst = torch.empty(B, T // 128, 64, 2, device=dev); dwh = torch.zeros(3, 64, 5, device=dev); dbh = torch.zeros(3, device=dev)
    def head():
        assert lib.tq_head_conv_bwd_ws(p(dpred), p(co), p(hh), p(ga), p(gs), p(wh), p(Gh), p(gst), p(dwh), p(dbh), B, T, 64, 3, 5,
                                       p(wsb) if use_ws else None, wsb.numel() if use_ws else 0, stream()) == 0
    print(f"stem wgrad {med(stem):7.1f} | head bwd {med(head):7.1f}")
