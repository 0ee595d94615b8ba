import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from tqdne_amd import _lib, ops
from test_hip_bwd import cl, ncw, ref_slot_sums
C0, C1, Co, k, T, scale = 128, 64, 64, 5, 333, 3e-5
for T in (333, 256, 384, 200):
    g = torch.Generator().manual_seed(C0 + C1 + Co + k + T)
    B, Cin = 2, C0 + C1
    w = torch.randn(Co, Cin, k, generator=g) / math.sqrt(Cin * k)
    dy = torch.randn(B, Co, T, generator=g) * scale
    x = torch.randn(B, Cin, T, generator=g) + 0.3
    a, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
    u = (x * a[:, :, None] + sh[:, :, None]).double().requires_grad_(True)
    F.conv1d(F.silu(u), w.double(), None, padding=2).backward(dy.double())
    d = torch.device('cuda:0')
    g0, g1, st = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x[:, :C0]), x1=cl(x[:, C0:]), gscale=a.to(d), gshift=sh.to(d), silu=True, stats=True, split=C0, wfmt=_lib.TQ_WFMT_F16_MX6)
    ref = ref_slot_sums(u.grad.float(), x)
    err = (st.cpu() - ref).abs() / ref.abs().max()
    print('T', T, 'st shape', tuple(st.shape), 'max err', float(err.max()))
    bad = (err > 1e-3).nonzero()
    print(' bad entries', len(bad), bad[:12].tolist())
    for wf in (_lib.TQ_WFMT_BF16X3,):
        _, _, st2 = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x[:, :C0]), x1=cl(x[:, C0:]), gscale=a.to(d), gshift=sh.to(d), silu=True, stats=True, split=C0, wfmt=wf)
        print('  bf16x3 max err', float(((st2.cpu() - ref).abs() / ref.abs().max()).max()))
