import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tqdne_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T, H, D = 16, 512, 4, 64
qkv = torch.randn(B, T, 3 * H * D, generator=g).to(dev)
A, S2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
hx = torch.randn(16, 4096, 64, generator=g).to(dev); hw = (0.1 * torch.randn(3, 64, 5, generator=g)).to(dev); hb = torch.randn(3, generator=g).to(dev)
cx = torch.randn(16, 1024, 256, generator=g).to(dev); cw = (0.02 * torch.randn(256, 256, 5, generator=g)).to(dev)
c64 = torch.randn(16, 4096, 64, generator=g).to(dev); w64 = (0.05 * torch.randn(64, 64, 5, generator=g)).to(dev)
gs = torch.ones(16, 256, device=dev); gh = torch.zeros(16, 256, device=dev)
mm_a = torch.randn(2048, 2048, generator=g).to(dev); mm_b = torch.randn(2048, 2048, generator=g).to(dev)
mmh_a, mmh_b = mm_a.bfloat16(), mm_b.bfloat16()
aggressors = {
    "attention": lambda: ops.attention(qkv, H),
    "conv 256->256 bf16x3": lambda: ops.conv1d(cx, cw, None, wfmt=_lib.TQ_WFMT_BF16X3),
    "conv 256->256 f16+mx8": lambda: ops.conv1d(cx, cw, None, wfmt=_lib.TQ_WFMT_F16_MX8),
    "conv 256->256 f16+mx8 GN+SiLU": lambda: ops.conv1d(cx, cw, None, wfmt=_lib.TQ_WFMT_F16_MX8, gscale=gs, gshift=gh, silu=True),
    "conv 64->64 bf16x3": lambda: ops.conv1d(c64, w64, None),
    "torch fp32 matmul": lambda: mm_a @ mm_b,
    "torch bf16 matmul": lambda: mmh_a @ mmh_b,
    "torch exp": lambda: torch.exp(cx),
}
ref = ops.head_conv(hx, hw, hb).clone(); torch.cuda.synchronize()
for an, afn in aggressors.items():
    bad = 0
    for it in range(10):
        outs = []
        for k in range(8):
            with torch.cuda.stream(S2):
                afn()
            with torch.cuda.stream(A):
                outs.append(ops.head_conv(hx, hw, hb))
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, ref) for o in outs)
    print(f"aggressor {an}: head corrupted {bad} of 80")
