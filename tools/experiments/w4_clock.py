#!/usr/bin/env python3
"""In-kernel clock of the one-wave-per-SIMD conv kernel (conv1d_w4.hip), VERDICT r3 item 5: after >= 2 s of back-to-back launches on
random data, every workgroup's d s_memtime / d s_memrealtime x 100 MHz of the LAST launch, median over workgroups.
Needs a library built with TQDNE_BUILD_EXPERIMENTS=1 and -DTQ_W4_STAMP (and, for the MFMA / weight / LDS-read stream alone,
-DTQ_W4_ABL_NOCONV), given in TQDNE_HIP_LIB.  usage: TQDNE_CONV_W4=1 TQDNE_HIP_LIB=... python tools/experiments/w4_clock.py C0 C1 Cout T [B] [seconds]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tqdne_amd import _lib, ops
C0, C1, Co, T = map(int, sys.argv[1:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
secs = float(sys.argv[6]) if len(sys.argv) > 6 else 2.5
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
assert run() == 0
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:          # >= 2 s of back-to-back launches: the clock has settled under this load
    for _ in range(200): run()
    torch.cuda.synchronize(); n += 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 20
rd = C.CDLL(_lib.lib_path()).tq_debug_read_w4_stamps
nwg = 4096   # (entries of workgroups that do not exist stay zero and are filtered below)
buf = (C.c_ulonglong * (4 * nwg))()
assert rd(buf, nwg) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
dt_clk, dt_real = a[:, 1] - a[:, 0], a[:, 3] - a[:, 2]
ok = (dt_real > 0) & (dt_clk > 0)
ghz = dt_clk[ok] / dt_real[ok] * 0.1
life_us = dt_real[ok] / 100.0
print(f"{os.path.basename(_lib.lib_path())}: {C0}+{C1}->{Co} T{T} B{B}: {us:.1f} us per launch after {n} warm launches ({secs:.1f} s); "
      f"in-kernel clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}) over {int(ok.sum())} workgroups; "
      f"workgroup life median {np.median(life_us):.1f} us")
