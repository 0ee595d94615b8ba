#!/usr/bin/env python3
"""Numerical emulation (numpy, CPU) of the contraction schemes considered for the fp32 conv product on the gfx950 matrix cores:
error of  y = sum_k x_k w_k  (K = 5 taps x 256 channels) against float64, on activations shaped like the network's
(SiLU(GroupNorm(.)) with a few outliers) and weights ~ N(0, 1/fan_in).  Backs the precision table of DESIGN.md section 3 and the
"next step" there (fp6 corrections at half the correction cost).  usage: python tools/emulate_schemes.py [seed]"""
import sys
import numpy as np

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def minifloat(v, ebits, mbits, bias, vmax):
    """round-to-nearest-even to a sign/exponent/mantissa format with subnormals, saturating at +-vmax"""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    emin = 1 - bias
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, emin)
    q = 2.0 ** (e - mbits)
    r = np.round(a / q) * q  # numpy rounds half to even
    r = np.minimum(r, vmax)
    return np.sign(v) * r


fp8 = lambda v: minifloat(v, 4, 3, 7, 448.0)        # OCP e4m3
e2m3 = lambda v: minifloat(v, 2, 3, 1, 7.5)          # OCP fp6
e3m2 = lambda v: minifloat(v, 3, 2, 3, 28.0)
f16 = lambda v: np.asarray(v, np.float32).astype(np.float16).astype(np.float64)


def bf16(v):
    u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def block_scale(v, fmt_max, block=32):
    """E8M0 scale per `block` consecutive elements of the last axis so that the block maximum fits the format"""
    s = v.reshape(*v.shape[:-1], -1, block)
    m = np.abs(s).max(-1, keepdims=True)
    sc = 2.0 ** np.ceil(np.log2(np.maximum(m, 1e-30) / fmt_max))
    return np.broadcast_to(sc, s.shape).reshape(v.shape)


N, K = 4096, 1280
x = rng.standard_normal((N, K)) * (0.5 + rng.random((1, K)) * 1.5) + rng.standard_normal((1, K)) * 0.5
x = x / (1 + np.exp(-x))                                  # SiLU
x[rng.random(x.shape) < 1e-3] *= 6.0                      # outliers
w = rng.standard_normal((K, 64)) / np.sqrt(K)
x = x.astype(np.float32).astype(np.float64)
w = w.astype(np.float32).astype(np.float64)
ref = x @ w
scale = np.abs(ref).max()
err = lambda y: np.abs(y - ref).max() / scale
rms = lambda y: np.sqrt(np.mean((y - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))

out = []
xh, wh = bf16(x), bf16(w)
xl, wl = bf16(x - xh), bf16(w - wh)
out.append(("bf16x3 (3 bf16 products)", 3.0, xh @ wh + xh @ wl + xl @ wh))
xh, wh = f16(x), f16(w)
xl, wl = x - xh, w - wh
out.append(("fp16 only (1 product)", 1.0, xh @ wh))
out.append(("fp16 x 2 (xh*wh + xh*wl16)", 2.0, xh @ wh + xh @ f16(wl)))
S = 2.0 ** 12
out.append(("f16 + mx8 (built): fp8 corrections, uniform scales", 2.0,
            xh @ wh + (fp8(xl * S) @ fp8(w) + fp8(x) @ fp8(wl * S)) / S))
out.append(("f16 + one fp8 correction (x side only)", 1.5, xh @ wh + (fp8(xl * S) @ fp8(w)) / S))
# fp6 corrections: weights block-scaled per 32 channels (free: done at pack time), activations with ONE uniform scale (cheap staging)
wt = w.T  # (co, K): blocks along K
sw = block_scale(wt, 7.5)
w6 = (e2m3(wt / sw) * sw).T
swl = block_scale(wl.T, 7.5)
wl6 = (e2m3(wl.T / swl) * swl).T
for name, xfmt, xmax_scale in (("e3m2", e3m2, 2.0 ** 10), ("e2m3", e2m3, 2.0 ** 11)):
    xl6 = xfmt(xl * xmax_scale) / xmax_scale          # uniform scale: small residuals flush, large ones saturate
    x6 = xfmt(x * 2.0) / 2.0
    out.append((f"f16 + fp6 corrections: w e2m3 block-scaled, x {name} uniform scale", 1.5, xh @ wh + xl6 @ w6 + x6 @ wl6))
sx = block_scale(xl, 7.5)
sx2 = block_scale(x, 7.5)
out.append(("f16 + fp6 corrections: both e2m3 block-scaled", 1.5,
            xh @ wh + (e2m3(xl / sx) * sx) @ w6 + (e2m3(x / sx2) * sx2) @ wl6))
print(f"{'scheme':78s} {'MFMA cost':>9s} {'max err':>9s} {'rms err':>9s}   (cost in bf16-product units per algorithmic product)")
for name, cost, y in out:
    print(f"{name:78s} {cost:9.2f} {err(y):9.2e} {rms(y):9.2e}")
