import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tqdne_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, T, C0, Co, K = 2, 384, 256, 256, 5
x0 = torch.randn(B, T, C0, generator=g).to(dev); w = (torch.randn(Co, C0, K, generator=g) / (K * C0) ** 0.5).to(dev)
bias = torch.randn(Co, generator=g).to(dev); gs = (torch.rand(B, C0, generator=g) + 0.5).to(dev); gh = torch.randn(B, C0, generator=g).to(dev)
ys = [ops.conv1d(x0, w, bias, gscale=gs, gshift=gh, silu=True, stats=True, wfmt=2)[0].clone() for _ in range(6)]
for i in range(1, 6):
    d = (ys[i] - ys[0]).abs().amax(dim=(0, 2))
    print("run", i, "equal to run 0:", torch.equal(ys[i], ys[0]), "rows differing:", (d > 0).nonzero().flatten().tolist()[:20])
