#!/usr/bin/env python3
"""Algorithmic FLOP and fused-minimum HBM bytes per waveform and UNet forward, by kernel class.
Byte model: every conv / attention core reads its input once and writes its output once in fp32; GroupNorm, SiLU,
dropout, bias / embedding / residual adds, concat and nearest up-sampling cost nothing (fused).  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
from tqdne_amd import UNetModel, paper_1d_unet_config, tiny_1d_unet_config

name = sys.argv[1] if len(sys.argv) > 1 else "paper"
cfg = paper_1d_unet_config() if name == "paper" else tiny_1d_unet_config()
T0 = 4096
m = UNetModel(**cfg)
fl, by = defaultdict(float), defaultdict(float)


def conv(cls, cin, cout, k, t_in, t_out, writes=True):
    fl[cls] += 2.0 * cin * cout * k * t_out
    by[cls] += 4.0 * (t_in * cin + (t_out * cout if writes else 0))


def res(rb, cin, T):
    co = rb.out_channels
    k = rb.in_layers[2].weight.shape[2]
    conv(f"ResBlock conv k={k}", cin, co, k, T, T)
    conv(f"ResBlock conv k={k}", co, co, k, T, T)
    if hasattr(rb.skip_connection, "weight"):
        conv("ResBlock 1x1 skip conv (fused into conv2: no extra output)", cin, co, 1, T, T, writes=False)
    return co


def attn(ab, T):
    C = ab.channels
    conv("attention qkv / proj 1x1 convs", C, 3 * C, 1, T, T)
    fl["attention core"] += 4.0 * C * T * T
    by["attention core"] += 4.0 * (3 * C * T + C * T)
    conv("attention qkv / proj 1x1 convs", C, C, 1, T, T)


stem = m.input_blocks[0][0]
conv("stem / head convs", stem.in_channels, stem.out_channels, stem.kernel_size[0], T0, T0)
T, hs, ch = T0, [stem.out_channels], stem.out_channels
for blk in list(m.input_blocks)[1:]:
    for layer in blk:
        kind = getattr(layer, "kind", None)
        if kind == "res":
            ch = res(layer, ch, T)
        elif kind == "attn":
            attn(layer, T)
        elif kind == "down":
            conv("down / up-sampling convs", ch, ch, layer.op.weight.shape[2], T, T // 2)
            T //= 2
    hs.append((ch, T))
hs[0] = (stem.out_channels, T0)
for layer in m.middle_block:
    kind = getattr(layer, "kind", None)
    if kind == "res":
        ch = res(layer, ch, T)
    elif kind == "attn":
        attn(layer, T)
for blk in m.output_blocks:
    sc, sT = hs.pop()
    assert sT == T
    cin = ch + sc
    for layer in blk:
        kind = getattr(layer, "kind", None)
        if kind == "res":
            ch = res(layer, cin, T)
        elif kind == "attn":
            attn(layer, T)
        elif kind == "up":
            conv("down / up-sampling convs", ch, ch, layer.conv.weight.shape[2], T, 2 * T)
            T *= 2
head = m.out[2]
conv("stem / head convs", head.in_channels, head.out_channels, head.kernel_size[0], T0, T0)
print(f"{name} UNet, per waveform (3 x {T0}) and forward:")
for k in sorted(fl, key=lambda k: -fl[k]):
    print(f"  {k:62s} {fl[k] / 1e9:7.2f} GFLOP  {by[k] / 1e6:7.1f} MB")
print(f"  {'total':62s} {sum(fl.values()) / 1e9:7.2f} GFLOP  {sum(by.values()) / 1e6:7.1f} MB   (weights: {sum(p.numel() for p in m.parameters()) * 4 / 1e6:.1f} MB per call)")
