import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import ops, _lib
dev = torch.device("cuda:0")
for (C, T) in ((256, 512), (256, 1024), (128, 2048)):
    x = torch.randn(64, T, C, device=dev); w = (torch.randn(C, C, 5, device=dev) / (5 * C) ** 0.5); b = torch.randn(C, device=dev)
    for wfmt in (0, 1):
        f = lambda: ops.conv1d(x, w, b, upsample=True, stats=True, wfmt=wfmt)
        for _ in range(3): f()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # time only the conv launch: pack happens inside ops.conv1d too, so measure with events around many calls and subtract pack cost separately
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"UPS C{C} T{T}->{2*T} wfmt {wfmt}: {e0.elapsed_time(e1)/20*1e3:.1f} us (incl. weight pack + output alloc)")
