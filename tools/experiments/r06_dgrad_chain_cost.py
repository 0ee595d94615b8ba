#!/usr/bin/env python3
"""Round 6 (verdict r5 item 4): what does the data gradient's chain epilogue cost per launch?  The same tq_conv1d_bwd_data launch of
representative layers of the paper UNet (B = 64), alone on its stream, 20 repetitions between two HIP events:
  plain      EPI = 1 with no chain: d x-hat written as it leaves the accumulators
  chain      x read back, x SiLU'(a x + s) x dropout mask, GroupNorm-backward slot sums (sum g, sum g x) emitted
and next to them the forward launch of the same conv (same multiply-adds, operands swapped) and the launch's byte floor
(dy + dx [+ x] in fp32 at 6.3 TB/s).  Which scheme each launch runs (fp16 + MX-fp6 or bf16x3) follows the plan's rule."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tqdne_amd import _lib, ops

dev = torch.device("cuda:0")
B = 64
lib = _lib.load()
# forward conv (C0 + C1 -> Co, k, T): its data gradient maps Co -> C0 + C1
LAYERS = [("input_blocks.8 conv1 (256 -> 256, T 1024)", 256, 0, 256, 5, 1024), ("output_blocks.3 conv1 (512 -> 256, T 1024)", 256, 256, 256, 5, 1024),
          ("output_blocks.6 conv1 (384 -> 128, T 2048)", 256, 128, 128, 5, 2048), ("output_blocks.8 conv1 (192 -> 128, T 2048)", 128, 64, 128, 5, 2048),
          ("output_blocks.9 conv1 (192 -> 64, T 4096)", 128, 64, 64, 5, 4096), ("output_blocks.11 conv1 (128 -> 64, T 4096)", 64, 64, 64, 5, 4096),
          ("input_blocks.1 conv1 (64 -> 64, T 4096)", 64, 0, 64, 5, 4096), ("output_blocks.3 skip (512 -> 256, k 1, T 1024)", 256, 256, 256, 1, 1024)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"{'layer':48s} {'scheme':7s} {'fwd us':>7s} {'plain':>7s} {'chain':>7s} {'+drop':>7s} {'chain cost':>10s} {'floor plain / chain us':>22s}")
for name, C0, C1, Co, K, T in LAYERS:
    Cin = C0 + C1
    g = torch.Generator(device="cpu").manual_seed(1)
    x0 = torch.randn(B, T, C0, generator=g).to(dev)
    x1 = torch.randn(B, T, C1, generator=g).to(dev) if C1 else None
    w = (torch.randn(Co, Cin, K, generator=g) / (K * Cin) ** 0.5).to(dev)
    dy = (torch.randn(B, T, Co, generator=g) * 1e-5).to(dev)
    a, s = (torch.rand(B, Cin, generator=g) + 0.5).to(dev), torch.randn(B, Cin, generator=g).to(dev)
    mx6 = Co % 64 == 0 and Cin % 128 == 0
    wfmt = _lib.TQ_WFMT_F16_MX6 if mx6 else _lib.TQ_WFMT_BF16X3
    amax = ops.amax_bits(dy) if mx6 else None
    g0 = torch.empty(B, T, C0, device=dev)
    g1 = torch.empty(B, T, C1, device=dev) if C1 else None
    # (the wrappers re-pack the weights per call: time the library call alone by building the descriptor once)
    wp = ops.pack_conv_weight(w, _lib.PACK_MODE_T[wfmt])
    st = torch.empty(B, (T + 127) // 128, Cin, 2, device=dev)

    def dgrad(flags, p=0.0):
        d = _lib.TqConvBwdDesc()
        d.B, d.T, d.C_dy, d.C_dx0, d.C_dx1, d.ktaps, d.flags = B, T, Co, C0, C1, K, flags
        d.dropout_site, d.dropout_p, d.dropout_seed = 3, p, 7
        d.wfmt = wfmt
        if mx6:
            d.dy_amax = amax.data_ptr()
        chain = flags != 0
        stream = torch.cuda.current_stream().cuda_stream
        pp = lambda t: None if t is None else t.data_ptr()

        def run():
            rc = lib.tq_conv1d_bwd_data(C.byref(d), pp(dy), pp(wp), pp(x0) if chain else None, pp(x1) if chain else None,
                                        pp(a) if chain else None, pp(s) if chain else None, pp(g0), pp(g1), pp(st) if chain else None, stream)
            assert rc == 0, rc
        return timed(run)

    t_plain = dgrad(0)
    t_chain = dgrad(_lib.TQ_BWD_GN | _lib.TQ_BWD_SILU | _lib.TQ_BWD_STATS)
    t_drop = dgrad(_lib.TQ_BWD_GN | _lib.TQ_BWD_SILU | _lib.TQ_BWD_STATS | _lib.TQ_BWD_DROPOUT, 0.1)
    # forward launch of the same conv through the pre-built descriptor path of tools/bench_conv.py
    fw = _lib.forward_wfmt(Co, [C0, C1])
    wpf = ops.pack_conv_weight(w, _lib.PACK_MODE[fw])
    y = torch.empty(B, T, Co, device=dev)
    stf = torch.empty(B, (T + 127) // 128, Co, 2, device=dev)
    fd = _lib.TqConvDesc()
    fd.B, fd.T_in, fd.T_out, fd.C_in0, fd.C_in1, fd.C_out = B, T, T, C0, C1, Co
    fd.ktaps, fd.stride, fd.pad, fd.upsample, fd.wfmt = K, 1, K // 2, 0, fw
    fd.flags = (3 if K == 5 else 0) | 16
    stream = torch.cuda.current_stream().cuda_stream
    pp = lambda t: None if t is None else t.data_ptr()

    def frun():
        rc = lib.tq_conv1d_fwd(C.byref(fd), pp(x0), pp(x1), pp(a) if K == 5 else None, pp(s) if K == 5 else None, pp(wpf), None, None, None, pp(y), pp(stf), stream)
        assert rc == 0, rc
    t_fwd = timed(frun)
    fl_plain = 4.0 * B * T * (Co + Cin) / 6.3e6
    fl_chain = 4.0 * B * T * (Co + 2 * Cin) / 6.3e6
    print(f"{name:48s} {'f16mx6' if mx6 else 'bf16x3':7s} {t_fwd:7.1f} {t_plain:7.1f} {t_chain:7.1f} {t_drop:7.1f} {t_chain - t_plain:+10.1f} {fl_plain:11.1f} / {fl_chain:6.1f}")
