import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T, C = 16, 4096, 64
x = torch.randn(B, T, C, generator=g).to(dev)
w = (0.1 * torch.randn(3, C, 5, generator=g)).to(dev)
b = torch.randn(3, generator=g).to(dev)
gs = (1 + 0.1 * torch.randn(B, C, generator=g)).to(dev); gh = (0.1 * torch.randn(B, C, generator=g)).to(dev)
co = torch.rand(B, generator=g).to(dev); ck = torch.rand(B, generator=g).to(dev)
src = torch.randn(B, 3, T, generator=g).to(dev)
ref = ops.head_conv(x, w, b, gs, gh, co, ck, src).clone()
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(4)]
xb = torch.randn(64, 1024, 256, generator=g).to(dev); wb = (0.02 * torch.randn(256, 256, 5, generator=g)).to(dev)
for mode in ("head x4 streams", "head + big conv on another stream", "head alone repeated"):
    bad = 0
    for it in range(30):
        outs = []
        if mode == "head x4 streams":
            for s in streams:
                with torch.cuda.stream(s):
                    for _ in range(3):
                        outs.append(ops.head_conv(x, w, b, gs, gh, co, ck, src))
        elif mode.startswith("head + big"):
            with torch.cuda.stream(streams[0]):
                for _ in range(4):
                    ops.conv1d(xb, wb, None)
            with torch.cuda.stream(streams[1]):
                for _ in range(12):
                    outs.append(ops.head_conv(x, w, b, gs, gh, co, ck, src))
        else:
            for _ in range(12):
                outs.append(ops.head_conv(x, w, b, gs, gh, co, ck, src))
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, ref) for o in outs)
    print(f"{mode}: {bad} mismatching outputs of {30 * 12}")
