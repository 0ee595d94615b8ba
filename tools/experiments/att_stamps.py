#!/usr/bin/env python3
"""Per-workgroup phase timeline of the attention forward kernel (needs the -DTQ_STAMP build: tqdne_amd/lib/stamp.so)."""
import os, sys, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TQDNE_HIP_LIB"] = os.path.join(root, "tqdne_amd", "lib", "stamp.so")
sys.path.insert(0, root)
import numpy as np, torch
from tqdne_amd import ops, _lib
B, T, H, D = (list(map(int, sys.argv[1:])) + [64, 512, 4, 64][len(sys.argv) - 1:])[:4]
lib = _lib.load()
qkv = torch.randn(B, T, 3 * H * D, device="cuda:0")
for _ in range(3):
    ops.attention(qkv, H)
torch.cuda.synchronize()
NW = min(4096, B * H * ((T + 127) // 128))
tl = (C.c_ulonglong * (8 * NW))()
lib.tq_debug_read_att_timeline.restype = C.c_int
lib.tq_debug_read_att_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.tq_debug_read_att_timeline(tl, NW) == 0
a = np.array(tl, dtype=np.float64).reshape(NW, 8)
r = a[:, :6] * 10.0
t0 = r[:, 0].min()
print(f"{NW} workgroups, first entry -> last exit {(r[:, 5].max() - t0) / 1e3:.1f} us; entry median {np.median(r[:, 0] - t0) / 1e3:.1f} us, "
      f"late entries (> 5 us) {(r[:, 0] - t0 > 5e3).mean():.2f}")
for nm, i0, i1 in (("Q load+split", 0, 1), ("K/V prologue", 1, 2), ("loop", 2, 3), ("last tile", 3, 4), ("store", 4, 5), ("life", 0, 5)):
    d = (r[:, i1] - r[:, i0]) / 1e3
    print(f"  {nm:13s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p95 {np.percentile(d, 95):7.2f}")
clk = (a[:, 7] - a[:, 6]) / np.maximum(1.0, (r[:, 5] - r[:, 0]))
print(f"  in-kernel clock: median {np.median(clk):.3f} GHz")
