import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tqdne_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T, H, D = 16, 512, 4, 64
qkv = torch.randn(B, T, 3 * H * D, generator=g).to(dev)
A, S2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
hx = torch.randn(16, 4096, 64, generator=g).to(dev); hw = (0.1 * torch.randn(3, 64, 5, generator=g)).to(dev); hb = torch.randn(3, generator=g).to(dev)
hw1 = (0.1 * torch.randn(3, 64, 1, generator=g)).to(dev)
dout = torch.randn(B, T, H * D, generator=g).to(dev)
o_ref, lse = ops.attention(qkv, H, return_lse=True)
aggressors = {
    "attention fwd2 (workspace)": lambda: ops.attention(qkv, H),
    "attention gen-1 (no workspace)": lambda: ops.attention(qkv, H, workspace=False),
    "attention bwd": lambda: ops.attention_bwd(qkv, o_ref, dout, lse, H),
}
victims = {"head k5": lambda: ops.head_conv(hx, hw, hb), "head k1": lambda: ops.head_conv(hx, hw1, hb)}
torch.cuda.synchronize()
for an, afn in aggressors.items():
    for vn, vfn in victims.items():
        ref = vfn().clone(); torch.cuda.synchronize()
        bad = 0; first = None
        for it in range(10):
            outs = []
            for k in range(8):
                with torch.cuda.stream(S2):
                    afn()
                with torch.cuda.stream(A):
                    outs.append(vfn())
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    if first is None:
                        first = o.clone()
        print(f"{an} vs {vn}: {bad} of 80 corrupted")
        if first is not None:
            d = (first - ref).abs()
            idx = torch.nonzero(d > 0)
            bs = sorted(set(idx[:, 0].tolist())); ts = idx[:, 2]
            tiles = sorted(set((int(b_), int(t_) // 128) for b_, t_ in zip(idx[:, 0].tolist(), ts.tolist())))
            print(f"   {idx.shape[0]} elements in {len(tiles)} (b, tile) workgroups; first tiles {tiles[:6]}; within-tile positions of first tile:",
                  sorted(set(int(t_) % 128 for b_, t_ in zip(idx[:, 0].tolist(), ts.tolist()) if (int(b_), int(t_) // 128) == tiles[0]))[:40])
            b0, tl0 = tiles[0]
            sl = slice(tl0 * 128, tl0 * 128 + 128)
            print("   got - ref (channel 0, first 8 bad positions):", [(int(t_) % 128, round(float(first[b0, 0, t_] - ref[b0, 0, t_]), 4)) for t_ in sorted(set(ts[(idx[:, 0] == b0) & (ts // 128 == tl0)].tolist()))[:8]])
