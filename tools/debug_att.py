import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T, H, D = 16, 512, 4, 64
qkv = torch.randn(B, T, 3 * H * D, generator=g).to(dev)
A, S2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
# victims
hx = torch.randn(16, 4096, 64, generator=g).to(dev); hw = (0.1 * torch.randn(3, 64, 5, generator=g)).to(dev); hb = torch.randn(3, generator=g).to(dev)
ex = torch.randn(1 << 22, generator=g).to(dev)
cx = torch.randn(16, 1024, 256, generator=g).to(dev); cw = (0.02 * torch.randn(256, 256, 5, generator=g)).to(dev)
sx = torch.randn(16, 3, 4096, generator=g).to(dev); sw = torch.randn(64, 3, 5, generator=g).to(dev); sb = torch.randn(64, generator=g).to(dev)
victims = {
    "head_conv (dyn LDS 41 KB)": lambda: ops.head_conv(hx, hw, hb),
    "torch elementwise": lambda: ex * 2.0 + 1.0,
    "conv mx8 (dyn LDS 68 KB)": lambda: ops.conv1d(cx, cw, None)[0],
    "stem conv": lambda: ops.stem_conv(sx, sw, sb)[0],
    "torch softmax": lambda: torch.softmax(cx, dim=-1),
}
att_ref = ops.attention(qkv, H).clone()
torch.cuda.synchronize()
for name, fn in victims.items():
    ref = fn().clone(); torch.cuda.synchronize()
    bad = badatt = 0
    for it in range(20):
        outs, atts = [], []
        for k in range(8):
            with torch.cuda.stream(S2):
                atts.append(ops.attention(qkv, H))
            with torch.cuda.stream(A):
                outs.append(fn())
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, ref) for o in outs)
        badatt += sum(not torch.equal(a, att_ref) for a in atts)
    print(f"victim {name}: {bad} of 160 corrupted; attention outputs corrupted: {badatt} of 160")
