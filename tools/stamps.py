import os, sys, ctypes as C
os.environ["TQDNE_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tqdne_amd", "lib", "stamp.so")
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import _lib
lib = _lib.load()
lib.tq_debug_read_stamps.restype = C.c_int
lib.tq_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 8)()
args = sys.argv[1:]
sys.argv = ["bench_one.py"] + args
lib.tq_debug_read_stamps(out, 1)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_one.py"), run_name="__main__")
torch.cuda.synchronize()
lib.tq_debug_read_stamps(out, 0)
n = max(1, out[5])
names = ["stage_load", "mfma_phase", "stage_write", "barrier", "loop_total"]
tot = out[4] / n
for i, nm in enumerate(names):
    print(f"{nm:12s} {out[i]/n:12.0f} cycles/wave  {100*out[i]/max(1,out[4]):5.1f}%")
