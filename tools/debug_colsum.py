import sys, torch
sys.path.insert(0, "/root/repo")
from tqdne_amd import ops
d = torch.device("cuda:0")
for (B, C, T) in [(2, 32, 4096), (2, 32, 300), (2, 64, 4096), (3, 32, 2048), (2, 32, 1024)]:
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(B, T, C, generator=g)
    for rep in range(3):
        obc, oc = ops.colsum(dy.to(d))
        ref = dy.sum(1)
        print(B, C, T, float((obc.cpu() - ref).abs().max()), float((oc.cpu() - ref.sum(0)).abs().max()))
