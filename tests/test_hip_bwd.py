"""Backward-kernel parity (GPU): each gradient kernel against torch autograd of the same op chain on the CPU (fp32)."""

import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-4


def dev():
    return torch.device("cuda:0")


def cl(x):
    return x.permute(0, 2, 1).contiguous().to(dev())


def ncw(y):
    return y.permute(0, 2, 1).cpu()


def ref_slot_sums(a, b):
    B, C, T = a.shape
    ns = (T + 127) // 128
    out = torch.zeros(B, ns, C, 2)
    for s in range(ns):
        sa, sb = a[:, :, s * 128:(s + 1) * 128].double(), b[:, :, s * 128:(s + 1) * 128].double()
        out[:, s, :, 0] = sa.sum(-1)
        out[:, s, :, 1] = (sa * sb).sum(-1)
    return out


@pytest.mark.parametrize("cin,cout,k,T", [(64, 64, 5, 256), (128, 256, 5, 200), (256, 128, 3, 127), (32, 64, 1, 300), (512, 256, 1, 100)])
def test_dgrad_plain(cin, cout, k, T):
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(cin + cout + k + T)
    x = torch.randn(2, cin, T, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    dy = torch.randn(2, cout, T, generator=g)
    F.conv1d(x, w, None, padding=k // 2).backward(dy)
    g0, _, _ = ops.conv1d_bwd_data(cl(dy), w.to(dev()))
    assert rel_err(ncw(g0), x.grad) < TOL


def test_dgrad_activation_chain_concat_and_stats():
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(11)
    B, C0, C1, Co, T = 2, 128, 64, 128, 333
    x0, x1 = torch.randn(B, C0, T, generator=g), torch.randn(B, C1, T, generator=g) + 0.5
    a, s = torch.randn(B, C0 + C1, generator=g), torch.randn(B, C0 + C1, generator=g)
    w = torch.randn(Co, C0 + C1, 5, generator=g) / 30
    dy = torch.randn(B, Co, T, generator=g)
    x = torch.cat([x0, x1], 1)
    u = (x * a[:, :, None] + s[:, :, None]).requires_grad_(True)
    F.conv1d(F.silu(u), w, None, padding=2).backward(dy)
    d = dev()
    g0, g1, st = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x0), x1=cl(x1), gscale=a.to(d), gshift=s.to(d), silu=True,
                                     stats=True, split=C0)
    got = torch.cat([ncw(g0), ncw(g1)], 1)
    assert rel_err(got, u.grad) < TOL
    assert rel_err(st.cpu(), ref_slot_sums(u.grad, x)) < TOL
    # accumulate flag
    base0, base1 = torch.ones_like(g0), torch.ones_like(g1)
    ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x0), x1=cl(x1), gscale=a.to(d), gshift=s.to(d), silu=True, split=C0,
                        accumulate_into=(base0, base1))
    assert rel_err(torch.cat([ncw(base0), ncw(base1)], 1), u.grad + 1) < TOL


@pytest.mark.parametrize("cin,cout,k,T,B", [(64, 64, 5, 256, 3), (128, 256, 5, 200, 2), (256, 128, 3, 127, 2), (32, 64, 1, 300, 2),
                                            (512, 256, 1, 64, 2), (96, 32, 5, 130, 5), (256, 768, 1, 512, 2)])
def test_wgrad_plain(cin, cout, k, T, B):
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(cin + cout + k + T)
    x = torch.randn(B, cin, T, generator=g)
    w = (torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)).requires_grad_(True)
    dy = torch.randn(B, cout, T, generator=g)
    F.conv1d(x, w, None, padding=k // 2).backward(dy)
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x), w.shape)
    assert rel_err(dw.cpu(), w.grad) < TOL


def test_wgrad_fused_prologue_concat():
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(12)
    B, C0, C1, Co, T = 3, 128, 64, 128, 333
    x0, x1 = torch.randn(B, C0, T, generator=g), torch.randn(B, C1, T, generator=g)
    a, s = torch.randn(B, C0 + C1, generator=g), torch.randn(B, C0 + C1, generator=g)
    w = (torch.randn(Co, C0 + C1, 5, generator=g) / 30).requires_grad_(True)
    dy = torch.randn(B, Co, T, generator=g)
    xh = F.silu(torch.cat([x0, x1], 1) * a[:, :, None] + s[:, :, None])
    F.conv1d(xh, w, None, padding=2).backward(dy)
    d = dev()
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, x1=cl(x1), gscale=a.to(d), gshift=s.to(d), silu=True)
    assert rel_err(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("C0,C1,k,T,B", [(64, 0, 5, 4096, 2), (128, 64, 5, 333, 3), (64, 64, 3, 127, 2), (192, 0, 5, 65, 5), (64, 0, 5, 31, 1)])
def test_wgrad_64_output_channels_four_wave_form(C0, C1, k, T, B):
    """Round 6: convs with exactly 64 output channels (the paper UNet's T = 4096 level) take the four-wave form of the shared-tile
    kernel (2 output-channel blocks x 2 halves of a 64-channel input tile) instead of half-filling the 128-channel tile: with the
    fused prologue (GroupNorm scale / shift, SiLU, dropout), a concat source, ragged lengths; same numbers as the 128-channel form."""
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(C0 + C1 + k + T)
    Co = 64
    x0 = torch.randn(B, C0, T, generator=g)
    x1 = torch.randn(B, C1, T, generator=g) if C1 else None
    a, s = torch.randn(B, C0 + C1, generator=g), torch.randn(B, C0 + C1, generator=g)
    w = (torch.randn(Co, C0 + C1, k, generator=g) / math.sqrt((C0 + C1) * k)).requires_grad_(True)
    dy = torch.randn(B, Co, T, generator=g)
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    F.conv1d(F.silu(x * a[:, :, None] + s[:, :, None]), w, None, padding=k // 2).backward(dy)
    d = dev()
    kw = dict(x1=cl(x1) if C1 else None, gscale=a.to(d), gshift=s.to(d), silu=True)
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, **kw)
    assert rel_err(dw.cpu(), w.grad) < TOL
    # dropout in the prologue: the same mask in both forms (the element index does not depend on the tiling)
    dwd = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, dropout_p=0.25, dropout_seed=77, dropout_site=3, **kw)
    # with fused column sums the launch takes the four-wave kernel of 128 output channels: the reference form for the new one
    bc, c1 = torch.zeros(B, Co, device=d), torch.ones(Co, device=d)
    dw4 = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, colsum=(bc, c1), **kw)
    dwd4 = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, dropout_p=0.25, dropout_seed=77, dropout_site=3,
                                 colsum=(torch.zeros(B, Co, device=d), None), **kw)
    assert rel_err(dw4.cpu(), w.grad) < TOL
    assert rel_err(dw.cpu(), dw4.cpu()) < 1e-5 and rel_err(dwd.cpu(), dwd4.cpu()) < 1e-5
    assert float((dwd - dw).norm() / dw.norm()) > 1e-2    # (the mask did something)
    assert rel_err(bc.cpu(), dy.sum(-1)) < 1e-5 and rel_err(c1.cpu(), 1 + dy.sum((0, 2))) < 1e-5


def test_wgrad_with_fused_column_sums_on_a_shape_of_the_eight_wave_form():
    """a launch that carries column sums plans AND runs the four-wave kernel (the plan used to be made for the eight-wave form)"""
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(19)
    B, Ci, Co, T = 3, 128, 128, 200
    x = torch.randn(B, Ci, T, generator=g)
    w = (torch.randn(Co, Ci, 5, generator=g) / 25).requires_grad_(True)
    dy = torch.randn(B, Co, T, generator=g)
    F.conv1d(x, w, None, padding=2).backward(dy)
    d = dev()
    bc, c1 = torch.zeros(B, Co, device=d), torch.zeros(Co, device=d)
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x), w.shape, colsum=(bc, c1))
    assert rel_err(dw.cpu(), w.grad) < TOL
    assert rel_err(bc.cpu(), dy.sum(-1)) < 1e-5 and rel_err(c1.cpu(), dy.sum((0, 2))) < 1e-5


@pytest.mark.parametrize("C,T", [(64, 256), (128, 251), (32, 130)])
def test_downsample_backward(C, T):
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(2, C, T, generator=g, requires_grad=True)
    w = (torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)).requires_grad_(True)
    y = F.conv1d(x, w, None, stride=2, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dyz = ops.zero_stuff(cl(dy), T)
    g0, _, _ = ops.conv1d_bwd_data(dyz, w.detach().to(dev()))
    assert rel_err(ncw(g0), x.grad) < TOL
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x.detach()), w.shape, stride=2)
    assert rel_err(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("C,T,k", [(128, 64, 5), (64, 100, 3), (256, 127, 5)])
def test_upsample_backward(C, T, k):
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(2, C, T, generator=g, requires_grad=True)
    w = (torch.randn(C, C, k, generator=g) / math.sqrt(k * C)).requires_grad_(True)
    y = F.conv1d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    gup, _, _ = ops.conv1d_bwd_data(cl(dy), w.detach().to(dev()))
    dx = ops.pair_sum(gup)
    assert rel_err(ncw(dx), x.grad) < TOL
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x.detach()), w.shape, upsample=True)
    assert rel_err(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("Cin,C,T", [(128, 128, 64), (256, 256, 200), (256, 128, 127), (64, 64, 333)])
def test_upsample_backward_in_the_two_phase_form(Cin, C, T):
    """round 6: Upsample (nearest x2 + conv k = 5) differentiated as its two-phase k = 3 conv -- the output gradient (B, 2T, C) read as
    (B, T, 2C): data gradient = k = 3 data gradient with the two-phase weights (both schemes), weight gradient = k = 3 weight gradient
    folded onto the five taps (tq_upsample_poly_wgrad_fold) -- vs autograd through F.interpolate + conv1d"""
    import ctypes as Ct
    from tqdne_amd import _lib, ops
    g = torch.Generator().manual_seed(Cin + C + T)
    x = torch.randn(2, Cin, T, generator=g, requires_grad=True)
    w = (torch.randn(C, Cin, 5, generator=g) / math.sqrt(5 * Cin)).requires_grad_(True)
    y = F.conv1d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=2)
    dy = torch.randn(y.shape, generator=g) * 1e-4
    y.backward(dy)
    wd = w.detach()
    w2 = torch.empty(2 * C, Cin, 3)   # (engine.UNetEngine.repack: even outputs (w0+w1, w2+w3, w4), odd ones (w0, w1+w2, w3+w4))
    w2[:C, :, 0], w2[:C, :, 1], w2[:C, :, 2] = wd[:, :, 0] + wd[:, :, 1], wd[:, :, 2] + wd[:, :, 3], wd[:, :, 4]
    w2[C:, :, 0], w2[C:, :, 1], w2[C:, :, 2] = wd[:, :, 0], wd[:, :, 1] + wd[:, :, 2], wd[:, :, 3] + wd[:, :, 4]
    d = dev()
    dyv = cl(dy).reshape(2, T, 2 * C)     # (B, 2T, C) channels-last IS (B, T, 2C): row m = [row 2m | row 2m + 1]
    for wfmt in (_lib.TQ_WFMT_BF16X3, _lib.TQ_WFMT_F16_MX6):
        if wfmt == _lib.TQ_WFMT_F16_MX6 and Cin % 64:
            continue
        dx, _, _ = ops.conv1d_bwd_data(dyv, w2.to(d), wfmt=wfmt)
        assert rel_err(ncw(dx), x.grad) < TOL, wfmt
    dw2 = ops.conv1d_bwd_weight(dyv, cl(x.detach()), (2 * C, Cin, 3))
    dw = torch.empty(C, Cin, 5, device=d)
    assert _lib.load().tq_upsample_poly_wgrad_fold(dw2.data_ptr(), dw.data_ptr(), C, Cin, None) == 0
    assert rel_err(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("C0,C1,Co,T", [(64, 0, 64, 256), (128, 64, 128, 200), (64, 32, 64, 130)])
def test_groupnorm_silu_conv_full_backward(C0, C1, Co, T):
    """GN32 (over a virtual concat, groups may straddle) -> SiLU -> conv: dx, dgamma, dbeta, dW vs autograd."""
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(C0 + C1 + T)
    B = 2
    d = dev()
    x0 = (torch.randn(B, C0, T, generator=g) * 2 + 0.3)
    x1 = (torch.randn(B, C1, T, generator=g) - 1) if C1 else None
    gamma = (1 + 0.2 * torch.randn(C0 + C1, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C0 + C1, generator=g)).requires_grad_(True)
    w = (torch.randn(Co, C0 + C1, 5, generator=g) / 20).requires_grad_(True)
    dy = torch.randn(B, Co, T, generator=g)
    x = (torch.cat([x0, x1], 1) if C1 else x0).clone().requires_grad_(True)
    F.conv1d(F.silu(F.group_norm(x, 32, gamma, beta, 1e-5)), w, None, padding=2).backward(dy)
    # HIP: forward statistics come from the producers; emulate them with identity 1x1 convs
    def stats_of(t):
        eye = torch.zeros(t.shape[1], t.shape[1], 1)
        eye[torch.arange(t.shape[1]), torch.arange(t.shape[1]), 0] = 1
        return ops.conv1d(cl(t), eye.to(d), None)[1]
    s0 = stats_of(x0)
    s1 = stats_of(x1) if C1 else None
    gs, gh, mr = ops.gn_finalize(s0, C0, T, gamma.detach().to(d), beta.detach().to(d), s1, C1)
    g0, g1, st = ops.conv1d_bwd_data(cl(dy), w.detach().to(d), x0=cl(x0), x1=cl(x1) if C1 else None, gscale=gs, gshift=gh,
                                     silu=True, stats=True, split=C0)
    ca, cb, cc, dgam, dbet = ops.gn_bwd_finalize(st, mr, gamma.detach().to(d), T)
    dx0 = ops.gn_bwd_apply(g0, cl(x0), (ca, cb, cc), C0 + C1, 0)
    got = ncw(dx0)
    if C1:
        dx1 = ops.gn_bwd_apply(g1, cl(x1), (ca, cb, cc), C0 + C1, C0)
        got = torch.cat([got, ncw(dx1)], 1)
    assert rel_err(got, x.grad) < TOL
    assert rel_err(dgam.cpu(), gamma.grad) < TOL
    assert rel_err(dbet.cpu(), beta.grad) < TOL
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x0), w.shape, x1=cl(x1) if C1 else None, gscale=gs, gshift=gh, silu=True)
    assert rel_err(dw.cpu(), w.grad) < TOL


def test_colsum():
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(2)
    for C in (32, 64, 256, 768):
        dy = torch.randn(3, C, 300, generator=g)
        sc = torch.rand(3, generator=g)
        obc, oc = ops.colsum(cl(dy), bscale=sc.to(dev()))
        ref = dy.sum(-1) * sc[:, None]
        assert rel_err(obc.cpu(), ref) < 1e-5 and rel_err(oc.cpu(), ref.sum(0)) < 1e-5


@pytest.mark.parametrize("nch,T,ws", [(3, 300, True), (6, 300, True), (6, 4064, True), (6, 190, False), (5, 129, True), (4, 128, False)])
def test_stem_and_head_backward(nch, T, ws):
    """first / last conv at the NCW boundary: 3 channels (waveforms), 6 (the envelope representation of the reference's real data: the
    streaming head backward serves up to 8 output channels since round 5), with the two-stage sums through a workspace and with atomics"""
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(3 + nch + T)
    d = dev()
    B = 2
    # stem
    x = torch.randn(B, nch, T, generator=g)
    sc = torch.rand(B, generator=g) + 0.5
    w = (torch.randn(64, nch, 5, generator=g) / 4).requires_grad_(True)
    dy = torch.randn(B, 64, T, generator=g)
    F.conv1d(x * sc[:, None, None], w, None, padding=2).backward(dy)
    dw = ops.stem_conv_bwd_weight(cl(dy), x.to(d), w.shape, in_scale=sc.to(d), workspace=ws)
    assert rel_err(dw.cpu(), w.grad) < 1e-5
    # head
    h = torch.randn(B, 64, T, generator=g)
    a, s = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    wh = (torch.randn(nch, 64, 5, generator=g) / 10).requires_grad_(True)
    bh = torch.randn(nch, generator=g).requires_grad_(True)
    co = torch.rand(B, generator=g) + 0.5
    dpred = torch.randn(B, nch, T, generator=g)
    u = (h * a[:, :, None] + s[:, :, None]).requires_grad_(True)
    (F.conv1d(F.silu(u), wh, bh, padding=2) * co[:, None, None]).backward(dpred)
    G, st, dwh, dbh = ops.head_conv_bwd(dpred.to(d), cl(h), wh.detach().to(d), a.to(d), s.to(d), co.to(d), workspace=ws)
    assert rel_err(ncw(G), u.grad) < 1e-5
    assert rel_err(dwh.cpu(), wh.grad) < 1e-5 and rel_err(dbh.cpu(), bh.grad) < 1e-5
    assert rel_err(st.cpu(), ref_slot_sums(u.grad, h)) < 1e-5


def test_dropout_forward_backward_consistency():
    """the mask regenerated in dgrad/wgrad equals the forward mask"""
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(4)
    d = dev()
    B, C, T = 2, 64, 256
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(C, C, 5, generator=g) / 18
    dy = torch.randn(B, C, T, generator=g)
    kw = dict(silu=True, dropout_p=0.25, dropout_seed=99, dropout_site=7)
    ones, zeros = torch.ones(B, C, device=d), torch.zeros(B, C, device=d)
    eye = torch.zeros(C, C, 1); eye[torch.arange(C), torch.arange(C), 0] = 1
    big = torch.full((B, T, C), 30.0, device=d)  # silu(30) == 30 in fp32: the identity conv returns 30 * mask / (1-p)
    mask = ncw(ops.conv1d(big, eye.to(d), None, gscale=ones, gshift=zeros, **kw)[0]) / 30.0
    y, _ = ops.conv1d(cl(x), w.to(d), None, gscale=ones, gshift=zeros, **kw)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv1d(F.silu(xr) * mask, wr, None, padding=2)
    assert rel_err(ncw(y), yr) < TOL
    yr.backward(dy)
    g0, _, _ = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x), gscale=ones, gshift=zeros, **kw)
    assert rel_err(ncw(g0), xr.grad) < TOL
    dw = ops.conv1d_bwd_weight(cl(dy), cl(x), w.shape, gscale=ones, gshift=zeros, **kw)
    assert rel_err(dw.cpu(), wr.grad) < TOL


@pytest.mark.parametrize("H,D,T", [(4, 64, 512), (2, 32, 62), (1, 128, 200), (2, 64, 127), (2, 64, 200), (1, 32, 320), (2, 64, 3), (2, 32, 65)])
def test_attention_backward(H, D, T):
    from tqdne_amd import ops
    g = torch.Generator().manual_seed(H * D + T)
    B = 2
    qkv = (torch.randn(B, 3 * H * D, T, generator=g) * 1.2).requires_grad_(True)
    dout = torch.randn(B, H * D, T, generator=g)
    q, k, v = qkv.chunk(3, dim=1)
    sc = 1 / math.sqrt(math.sqrt(D))
    w = torch.einsum("bct,bcs->bts", (q * sc).reshape(B * H, D, T), (k * sc).reshape(B * H, D, T))
    w = torch.softmax(w.float(), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v.reshape(B * H, D, T)).reshape(B, -1, T)
    ref.backward(dout)
    x = cl(qkv.detach())
    out, lse = ops.attention(x, H, return_lse=True)
    assert rel_err(ncw(out), ref) < TOL
    for ws in (True, False):   # second-generation kernels (planes in a workspace; D = 128 falls through) | first generation
        got = ncw(ops.attention_bwd(x, out, cl(dout), lse, H, workspace=ws))
        for name, sl in (("dq", slice(0, H * D)), ("dk", slice(H * D, 2 * H * D)), ("dv", slice(2 * H * D, 3 * H * D))):
            assert rel_err(got[:, sl], qkv.grad[:, sl]) < TOL, (name, ws)


@pytest.mark.parametrize("H,D,T", [(4, 64, 512), (2, 32, 62), (2, 64, 127)])
def test_attention_backward_on_the_forwards_kv_planes_is_bit_identical(H, D, T):
    """tq_attention_bwd_ws_kv (round 5): the backward takes the K / V planes the training forward wrote into its workspace instead of
    re-deriving them from qkv -- same planes, same kernels behind the prep pass: dqkv and delta to the bit."""
    from tqdne_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 * H + D + T)
    B = 2
    d = dev()
    qkv = (torch.randn(B, T, 3 * H * D, generator=g) * 1.2).to(d)
    dout = torch.randn(B, T, H * D, generator=g).to(d)
    out, lse = torch.empty(B, T, H * D, device=d), torch.empty(B, H, T, device=d)
    nws = lib.tq_attention_workspace_bytes(B, T, H, D)
    kv = torch.zeros(nws, dtype=torch.uint8, device=d)   # (padding rows t >= T must be zero)
    p = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tq_attention_fwd(p(qkv), p(out), p(lse), p(kv), B, T, H, D, st) == 0
    res = []
    for use_kv in (False, True):
        dqkv, delta = torch.full_like(qkv, float("nan")), torch.empty(B, H, T, device=d)
        ws = torch.full((2 * nws,), 0xFF, dtype=torch.uint8, device=d)   # (garbage: what the kv route does not write it must not read)
        if use_kv:
            rc = lib.tq_attention_bwd_ws_kv(p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), p(ws), p(kv), B, T, H, D, st)
        else:
            rc = lib.tq_attention_bwd_ws(p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), p(ws), B, T, H, D, st)
        assert rc == 0
        torch.cuda.synchronize()
        res.append((dqkv, delta))
    assert torch.isfinite(res[1][0]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert lib.tq_attention_bwd_ws_kv(p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), p(ws), None, B, T, H, D, st) == -1   # TQ_ERR_ARG
    assert lib.tq_attention_bwd_ws_kv(p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), p(ws), p(kv), B, T, H, 128, st) == -2  # TQ_ERR_SHAPE


# ---------------------------------------------------------------------------------------------------------------- round 4
@pytest.mark.parametrize("cin,cout,k,T,scale", [(128, 256, 5, 200, 1e-6), (256, 256, 5, 333, 3e-5), (256, 128, 3, 127, 1.0),
                                                (512, 256, 1, 100, 1e-9), (384, 256, 5, 130, 2e3), (128, 768, 1, 256, 1e-6)])
def test_dgrad_f16_mx6_on_gradient_scale_inputs(cin, cout, k, T, scale):
    """the fp16 + MX-fp6 data gradient: dy at the magnitudes gradients have (1e-9 ... 1e3, far outside fp16's normal range unscaled),
    heavy-tailed, scaled inside the kernel by the power of two that tq_colsum's amax output selects -- vs fp64"""
    from tqdne_amd import _lib, ops
    g = torch.Generator().manual_seed(cin + cout + k + T)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    dy = torch.randn(2, cout, T, generator=g)
    dy = torch.where(torch.rand(dy.shape, generator=g) < 0.02, dy * 30.0, dy) * scale     # heavy tails
    ref = F.conv_transpose1d(dy.double(), w.double(), padding=k // 2)
    g0, _, _ = ops.conv1d_bwd_data(cl(dy), w.to(dev()), wfmt=_lib.TQ_WFMT_F16_MX6)
    e = rel_err(ncw(g0), ref)
    gb, _, _ = ops.conv1d_bwd_data(cl(dy), w.to(dev()))
    eb = rel_err(ncw(gb), ref)
    print(f"dgrad {cin}->{cout} k{k} at |dy| ~ {scale:g}: f16+mx6 {e:.2e}, bf16x3 {eb:.2e}")
    assert e < TOL


def test_dgrad_f16_mx6_chain_concat_stats_and_accumulate():
    from tqdne_amd import _lib, ops
    g = torch.Generator().manual_seed(21)
    B, C0, C1, Co, T = 2, 256, 128, 128, 333
    x0, x1 = torch.randn(B, C0, T, generator=g), torch.randn(B, C1, T, generator=g) + 0.5
    a, s = torch.randn(B, C0 + C1, generator=g), torch.randn(B, C0 + C1, generator=g)
    w = torch.randn(Co, C0 + C1, 5, generator=g) / 30
    dy = torch.randn(B, Co, T, generator=g) * 1e-5
    x = torch.cat([x0, x1], 1)
    u = (x * a[:, :, None] + s[:, :, None]).double().requires_grad_(True)
    F.conv1d(F.silu(u), w.double(), None, padding=2).backward(dy.double())
    d = dev()
    g0, g1, st = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x0), x1=cl(x1), gscale=a.to(d), gshift=s.to(d), silu=True,
                                     stats=True, split=C0, wfmt=_lib.TQ_WFMT_F16_MX6)
    got = torch.cat([ncw(g0), ncw(g1)], 1)
    assert rel_err(got, u.grad) < TOL
    assert rel_err(st.cpu(), ref_slot_sums(u.grad.float(), x)) < TOL
    base0, base1 = torch.full_like(g0, 1e-5), torch.full_like(g1, 1e-5)
    ops.conv1d_bwd_data(cl(dy), w.to(d), x0=cl(x0), x1=cl(x1), gscale=a.to(d), gshift=s.to(d), silu=True, split=C0,
                        accumulate_into=(base0, base1), wfmt=_lib.TQ_WFMT_F16_MX6)
    assert rel_err(torch.cat([ncw(base0), ncw(base1)], 1), u.grad + 1e-5) < TOL


@pytest.mark.parametrize("C0,C1,Co,k,T,scale", [(64, 0, 64, 5, 4096, 1e-6), (128, 64, 64, 5, 333, 3e-5), (64, 0, 128, 3, 200, 1.0),
                                                 (192, 0, 64, 1, 130, 1e-7), (64, 0, 256, 5, 127, 1e-4)])
def test_dgrad_f16_mx6_64_channel_tile(C0, C1, Co, k, T, scale):
    """round 6: data gradients whose produced channel count is a multiple of 64 but not of 128 (the 64- and 192-channel inputs of the
    T = 4096 level) in the fp16 + MX-fp6 scheme on the 64-channel x 128-position tile: plain, and with the GN + SiLU chain, the
    statistics (two 64-position waves share a slot: LDS hand-over) and the split over two concat sources -- vs fp64"""
    from tqdne_amd import _lib, ops
    g = torch.Generator().manual_seed(C0 + C1 + Co + k + T)
    B, Cin = 2, C0 + C1
    w = torch.randn(Co, Cin, k, generator=g) / math.sqrt(Cin * k)
    dy = torch.randn(B, Co, T, generator=g)
    dy = torch.where(torch.rand(dy.shape, generator=g) < 0.02, dy * 30.0, dy) * scale
    d = dev()
    ref = F.conv_transpose1d(dy.double(), w.double(), padding=k // 2)
    g0, g1, _ = ops.conv1d_bwd_data(cl(dy), w.to(d), split=C0 if C1 else None, wfmt=_lib.TQ_WFMT_F16_MX6)
    got = torch.cat([ncw(g0), ncw(g1)], 1) if C1 else ncw(g0)
    e = rel_err(got, ref)
    print(f"dgrad (64-channel tile) {Co} -> {C0}+{C1} k{k} T={T} at |dy| ~ {scale:g}: {e:.2e}")
    assert e < TOL
    if k == 5:
        x = torch.randn(B, Cin, T, generator=g) + 0.3
        a, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
        u = (x * a[:, :, None] + sh[:, :, None]).double().requires_grad_(True)
        F.conv1d(F.silu(u), w.double(), None, padding=2).backward(dy.double())
        xs = (cl(x[:, :C0]), cl(x[:, C0:]) if C1 else None)
        g0, g1, st = ops.conv1d_bwd_data(cl(dy), w.to(d), x0=xs[0], x1=xs[1], gscale=a.to(d), gshift=sh.to(d), silu=True, stats=True,
                                         split=C0 if C1 else None, wfmt=_lib.TQ_WFMT_F16_MX6)
        got = torch.cat([ncw(g0), ncw(g1)], 1) if C1 else ncw(g0)
        assert rel_err(got, u.grad) < TOL
        assert rel_err(st.cpu(), ref_slot_sums(u.grad.float(), x)) < TOL


def test_dgrad_f16_mx6_refuses_unsupported_shapes_and_missing_amax():
    import ctypes as C
    from tqdne_amd import _lib, ops
    lib = _lib.load()
    d = _lib.TqConvBwdDesc()
    d.B, d.T, d.C_dy, d.C_dx0, d.C_dx1, d.ktaps, d.wfmt = 1, 64, 64, 128, 0, 5, _lib.TQ_WFMT_F16_MX6
    x = torch.zeros(64 * 128, device=dev())
    assert lib.tq_conv1d_bwd_data(C.byref(d), x.data_ptr(), x.data_ptr(), None, None, None, None, x.data_ptr(), None, None, None) == -1  # no amax
    am = torch.zeros(_lib.TQ_AMAX_WORDS, dtype=torch.int32, device=dev())
    d.dy_amax = am.data_ptr()
    d.C_dx0 = 96      # 64 does not divide the produced channels (round 6: 64 | C_dx is served by the 64-channel tile)
    assert lib.tq_conv1d_bwd_data(C.byref(d), x.data_ptr(), x.data_ptr(), None, None, None, None, x.data_ptr(), None, None, None) == -2
    d.C_dx0, d.C_dy = 128, 32
    assert lib.tq_conv1d_bwd_data(C.byref(d), x.data_ptr(), x.data_ptr(), None, None, None, None, x.data_ptr(), None, None, None) == -2


@pytest.mark.parametrize("C,T,B", [(64, 4096, 2), (256, 300, 3), (192, 130, 2), (512, 64, 2)])
def test_gn_bwd_apply_with_fused_column_sums_and_amax(C, T, B):
    """tq_gn_bwd_apply_colsum = tq_gn_bwd_apply followed by tq_colsum of its output (bit-identical dx; sums to rounding; exact max)"""
    from tqdne_amd import _lib, ops
    g = torch.Generator().manual_seed(C + T)
    d = dev()
    G, x, r = (torch.randn(B, T, C, generator=g).to(d) for _ in range(3))
    coefs = tuple(torch.randn(B, C, generator=g).to(d) for _ in range(3))
    ref = ops.gn_bwd_apply(G, x, coefs, C, r=r)
    am = torch.zeros(_lib.TQ_AMAX_WORDS, dtype=torch.int32, device=d)
    dx, obc, oc = ops.gn_bwd_apply_colsum(G, x, coefs, C, r=r, amax=am)
    assert torch.equal(dx, ref)
    assert rel_err(obc.cpu(), ref.double().sum(1).cpu()) < 1e-5 and rel_err(oc.cpu(), ref.double().sum((0, 1)).cpu()) < 1e-5
    assert ops.amax_value(am) == float(ref.abs().max())
    # accumulate form on a source of a concat (coefficient offset), no residual
    C2 = C // 2
    G2, x2 = G[:, :, :C2].contiguous(), x[:, :, :C2].contiguous()
    base = torch.randn(B, T, C2, generator=g).to(d)
    ref2 = ops.gn_bwd_apply(G2, x2, coefs, C, c_offset=C - C2, accumulate_into=base.clone())
    dx2, obc2, _ = ops.gn_bwd_apply_colsum(G2, x2, coefs, C, c_offset=C - C2, accumulate_into=base.clone(), total=False)
    assert torch.equal(dx2, ref2) and rel_err(obc2.cpu(), ref2.double().sum(1).cpu()) < 1e-5
    # the stand-alone pass reports the same maximum; a NaN anywhere is carried as +inf
    am2 = torch.zeros(_lib.TQ_AMAX_WORDS, dtype=torch.int32, device=d)
    ops.colsum(ref, amax=am2)
    assert ops.amax_value(am2) == ops.amax_value(am)
    bad = ref.clone(); bad[B - 1, T // 2, C - 3] = float("nan")
    assert math.isinf(ops.amax_value(ops.amax_bits(bad)))
