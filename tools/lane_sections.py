#!/usr/bin/env python3
"""Which part of the UNet forward gains from sampler lanes?  Times the launch lists of the high-resolution (T >= 2048) and the
low-resolution (T <= 1024) sections separately, as ONE B = 64 launch list on one stream and as 4 lanes of B = 16 on 4 streams
(numerics are garbage when a section runs alone; timing is what counts).  Developer probe, GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import UNetModel, paper_1d_unet_config, engine as E
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = UNetModel(**paper_1d_unet_config()).to(dev).eval()
with torch.no_grad():
    for p in m.parameters():
        if torch.count_nonzero(p) == 0:
            p.normal_(0, 0.02)
B, T = 64, 4096
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, T, generator=g).to(dev); t = torch.rand(B, generator=g).to(dev); c = torch.randn(B, 5, generator=g).to(dev)
full = m._engine(B, T, dev)
lanes = [m._engine(B // 4, T, dev, lane=i) for i in range(4)]
with torch.no_grad():
    full.forward(x, t, c, infer=True)
    for i, e in enumerate(lanes):
        e.forward(x[16 * i:16 * i + 16].contiguous(), t[16 * i:16 * i + 16].contiguous(), c[16 * i:16 * i + 16].contiguous(), infer=True)
torch.cuda.synchronize()
names = [op[2] for op in full.ops_infer]
def idx(prefix, first=True):
    hits = [i for i, n in enumerate(names) if prefix in n]
    return hits[0] if first else hits[-1]
lo0 = idx("conv:input_blocks.6.0.op") + 1          # everything behind the second down-sampling conv: T <= 1024
lo1 = idx("conv:output_blocks.5.1.conv")           # ... up to (not including) the up-sampling conv back to T = 2048
sections = {"high-res head (T >= 2048, down path)": (0, lo0), "low-res middle (T <= 1024)": (lo0, lo1), "high-res tail (T >= 2048, up path)": (lo1, len(names))}
streams = [torch.cuda.current_stream(dev)] + [E.side_stream(dev, i) for i in range(1, 4)]
def run(eng, a, b, stream):
    for fn, args, what, _ in eng.ops_infer[a:b]:
        rc = fn(*args, stream.cuda_stream)
        assert rc == 0, (what, rc)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
main = streams[0]
tot_f = tot_l = 0
for name, (a, b) in sections.items():
    def f_full():
        run(full, a, b, main)
    def f_lanes():
        for s in streams[1:]:
            s.wait_stream(main)
        for e, s in zip(lanes, streams):
            run(e, a, b, s)
        for s in streams[1:]:
            main.wait_stream(s)
    tf, tl = timeit(f_full), timeit(f_lanes)
    tot_f += tf; tot_l += tl
    print(f"{name:42s} ops {a:3d}..{b:3d}: one B=64 list {tf:7.0f} us   4 lanes x B=16 {tl:7.0f} us   ratio {tl / tf:.3f}")
print(f"{'sum of sections':42s}              one B=64 list {tot_f:7.0f} us   4 lanes x B=16 {tot_l:7.0f} us")
def whole_full():
    run(full, 0, len(names), main)
def whole_lanes():
    for s in streams[1:]:
        s.wait_stream(main)
    for e, s in zip(lanes, streams):
        run(e, 0, len(names), s)
    for s in streams[1:]:
        main.wait_stream(s)
print(f"{'whole list':42s}              one B=64 list {timeit(whole_full):7.0f} us   4 lanes x B=16 {timeit(whole_lanes):7.0f} us")
