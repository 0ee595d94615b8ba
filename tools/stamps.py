import os, sys, ctypes as C
os.environ["TQDNE_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tqdne_amd", "lib", "stamp.so")
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import _lib
lib = _lib.load()
lib.tq_debug_read_stamps.restype = C.c_int
lib.tq_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 12)()
args = sys.argv[1:]
sys.argv = ["bench_one.py"] + args
lib.tq_debug_read_stamps(out, 1)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_one.py"), run_name="__main__")
torch.cuda.synchronize()
lib.tq_debug_read_stamps(out, 0)
n = max(1, out[5])
names = ["stage_load", "mfma_phase", "stage_write", "barrier", "loop_total"]  # phase sums of workgroup 7
tot = out[4] / n
for i, nm in enumerate(names):
    if nm is None:
        continue
    print(f"{nm:12s} {out[i]/n:12.0f} cycles/wave  {100*out[i]/max(1,out[4]):5.1f}%")
import numpy as np
lib.tq_debug_read_timeline.restype = C.c_int
lib.tq_debug_read_timeline.argtypes = [C.c_void_p, C.c_int]
NW = int(os.environ.get("TQ_NWG", "512"))
tl = (C.c_ulonglong * (8 * NW))()
lib.tq_debug_read_timeline(tl, NW)
a = np.array(tl, dtype=np.float64).reshape(NW, 8)
r = a[:, :4] * 10.0  # ns (100 MHz)
t0 = r[:, 0].min()
print(f"last launch: {NW} workgroups, first entry -> last exit {(r[:, 3].max() - t0) / 1e3:.1f} us")
print(f"  entry times: median {np.median(r[:, 0] - t0) / 1e3:.1f} us, 2nd-round share {(r[:, 0] - t0 > 5e3).mean():.2f}")
for nm, i0, i1 in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3), ("life", 0, 3)):
    d = (r[:, i1] - r[:, i0]) / 1e3
    print(f"  {nm:9s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p95 {np.percentile(d, 95):7.2f}")
clk = (a[:, 7] - a[:, 4]) / np.maximum(1.0, (r[:, 3] - r[:, 0]))
print(f"  in-kernel clock (s_memtime / s_memrealtime): median {np.median(clk):.3f} GHz")
