"""Thin functional wrappers over the C ABI (allocate outputs, pack weights, launch on the current stream).
Used by the parity tests and handy for experiments; the engine binds the same entry points directly."""

from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (TQ_CONV_DROPOUT, TQ_CONV_EMB, TQ_CONV_GN, TQ_CONV_RES, TQ_CONV_SILU, TQ_CONV_STATS, STAT_SLOT,
                   TqConvDesc, check)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def nslots(T):
    return (T + STAT_SLOT - 1) // STAT_SLOT


def pack_conv_weight(w: torch.Tensor, mode: int = 0) -> torch.Tensor:
    lib = _lib.load()
    co, ci, k = w.shape
    out = torch.empty(lib.tq_conv_weight_pack_bytes(co, ci, k, mode), dtype=torch.uint8, device=w.device)
    check(lib.tq_pack_conv_weight(_p(w.contiguous()), co, ci, k, mode, _p(out), _stream(w.device)), "pack")
    return out


def conv1d(x0, weight, bias=None, *, x1=None, gscale=None, gshift=None, silu=False, emb=None, residual=None,
           stride=1, upsample=False, stats=True, dropout_p=0.0, dropout_seed=0, dropout_site=0, skip=None, wfmt=None, t_tile=0,
           gn_fold=None):
    """x0/x1 (B, T, C) channels-last fp32; weight (C_out, C_in, K) torch layout.  Returns (y, stats|None).
    skip=(sx0, sx1|None, w_skip (C_out, Cs, 1), b_skip|None): fused 1x1 conv of the un-activated sx (tq_conv1d_fwd_skip).
    t_tile=32: the small tile (TqConvDesc.t_tile); the statistics then have one slot per 32 positions.
    gn_fold=(stats0, stats1|None, slot0, slot1, gamma, beta, mean_rstd|None): the launch folds its own GroupNorm (TqConvDesc.gn_fold) and
    WRITES the coefficients into ``gscale`` / ``gshift`` (pass empty (B, C_in) tensors) and ``mean_rstd``."""
    lib = _lib.load()
    B, T_in, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[2]
    C_out, C_in, K = weight.shape
    assert C_in == C0 + C1
    T_out = (T_in + 2 * (K // 2) - K) // 2 + 1 if stride == 2 else (2 * T_in if upsample else T_in)
    y = torch.empty(B, T_out, C_out, device=x0.device)
    st = torch.empty(B, (T_out + 31) // 32 if t_tile == 32 else nslots(T_out), C_out, 2, device=x0.device) if stats else None
    d = TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T_in, T_out, C0, C1, C_out
    d.ktaps, d.stride, d.pad, d.upsample = K, stride, K // 2, int(upsample)
    d.t_tile = t_tile
    f = 0
    if gscale is not None:
        f |= TQ_CONV_GN
    if silu:
        f |= TQ_CONV_SILU
    if emb is not None:
        f |= TQ_CONV_EMB
    if residual is not None:
        f |= TQ_CONV_RES
    if stats:
        f |= TQ_CONV_STATS
    if dropout_p > 0:
        f |= TQ_CONV_DROPOUT
    d.flags = f
    d.emb_stride = 0 if emb is None else emb.stride(0)
    d.dropout_site, d.dropout_p, d.dropout_seed = dropout_site, dropout_p, dropout_seed
    if wfmt is None:
        srcs = [C0, C1] + ([skip[0].shape[2], 0 if skip[1] is None else skip[1].shape[2]] if skip is not None else [])
        wfmt = _lib.forward_wfmt(C_out, srcs, stride, upsample, fused_skip=skip is not None,
                                 k5_act=K == 5 and gscale is not None and silu and stride == 1 and not upsample and t_tile == 0)
    d.wfmt = wfmt
    if gn_fold is not None:
        f = _lib.TqGnFold()
        f.stats0, f.stats1, f.slot0, f.slot1 = _p(gn_fold[0]), _p(gn_fold[1]), gn_fold[2], gn_fold[3]
        f.gamma, f.beta, f.mean_rstd = _p(gn_fold[4]), _p(gn_fold[5]), _p(gn_fold[6])
        d.gn_fold = C.pointer(f)
    pmode = _lib.PACK_MODE[wfmt]
    wp = pack_conv_weight(weight, pmode)
    if skip is not None:
        sx0, sx1, wsk, bsk = skip
        assert residual is None
        d.C_skip0, d.C_skip1 = sx0.shape[2], (0 if sx1 is None else sx1.shape[2])
        wp = torch.cat([wp, pack_conv_weight(wsk, pmode)])
        check(lib.tq_conv1d_fwd_skip(C.byref(d), _p(x0), _p(x1), _p(gscale), _p(gshift), _p(wp), _p(bias), _p(emb), _p(sx0),
                                     _p(sx1), _p(bsk), _p(y), _p(st), _stream(x0.device)), "conv1d_skip")
        return y, st
    check(lib.tq_conv1d_fwd(C.byref(d), _p(x0), _p(x1), _p(gscale), _p(gshift), _p(wp), _p(bias), _p(emb), _p(residual),
                            _p(y), _p(st), _stream(x0.device)), "conv1d")
    return y, st


def gn_finalize(stats0, C0, T, gamma, beta, stats1=None, C1=0, slot0=0, slot1=0):
    """slot0 / slot1: positions per statistics slot of each source (0 = 128; 32 for the output of a t_tile = 32 conv)"""
    lib = _lib.load()
    B = stats0.shape[0]
    Cn = C0 + C1
    gs, gh = torch.empty(B, Cn, device=gamma.device), torch.empty(B, Cn, device=gamma.device)
    mr = torch.empty(B, 32, 2, device=gamma.device)
    check(lib.tq_gn_finalize(_p(stats0), C0, _p(stats1), C1, B, T, _p(gamma), _p(beta), _p(gs), _p(gh), _p(mr), slot0, slot1,
                             _stream(gamma.device)), "gn_finalize")
    return gs, gh, mr


def stem_conv(x_nct, weight, bias, in_scale=None, stats=True):
    lib = _lib.load()
    B, Cin, T = x_nct.shape
    Cout, _, K = weight.shape
    y = torch.empty(B, T, Cout, device=x_nct.device)
    st = torch.empty(B, nslots(T), Cout, 2, device=x_nct.device) if stats else None
    check(lib.tq_stem_conv_fwd(_p(x_nct.contiguous()), _p(in_scale), _p(weight.contiguous()), _p(bias), _p(y), _p(st), B, Cin,
                               T, Cout, K, _stream(x_nct.device)), "stem")
    return y, st


def head_conv(x, weight, bias, gscale=None, gshift=None, c_out=None, c_skip=None, skip_src=None):
    lib = _lib.load()
    B, T, Cin = x.shape
    Cout, _, K = weight.shape
    y = torch.empty(B, Cout, T, device=x.device)
    check(lib.tq_head_conv_fwd(_p(x), _p(gscale), _p(gshift), _p(weight.contiguous()), _p(bias), _p(c_out), _p(c_skip),
                               _p(skip_src), _p(y), B, T, Cin, Cout, K, _stream(x.device)), "head")
    return y


def attention(qkv, heads, return_lse=False, workspace=True):
    lib = _lib.load()
    B, T, C3 = qkv.shape
    D = C3 // (3 * heads)
    out = torch.empty(B, T, heads * D, device=qkv.device)
    lse = torch.empty(B, heads, T, device=qkv.device) if return_lse else None
    ws = torch.empty(lib.tq_attention_workspace_bytes(B, T, heads, D), dtype=torch.uint8, device=qkv.device) if workspace else None
    check(lib.tq_attention_fwd(_p(qkv), _p(out), _p(lse), _p(ws), B, T, heads, D, _stream(qkv.device)), "attention")
    return (out, lse) if return_lse else out


def attention_bwd(qkv, out, dout, lse, heads, workspace=True):
    lib = _lib.load()
    B, T, C3 = qkv.shape
    D = C3 // (3 * heads)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, heads, T, device=qkv.device)
    # (the second-generation kernels exist for D = 32 / 64; D = 128 falls through to the first generation, which takes no scratch)
    ws = (torch.empty(2 * lib.tq_attention_workspace_bytes(B, T, heads, D), dtype=torch.uint8, device=qkv.device)
          if (workspace and D in (32, 64)) else None)
    check(lib.tq_attention_bwd_ws(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), _p(ws), B, T, heads, D,
                                  _stream(qkv.device)), "attention bwd")
    return dqkv


def linear(x, w, b=None):
    lib = _lib.load()
    B, E = x.shape
    N = w.shape[0]
    out = torch.empty(B, N, device=x.device)
    check(lib.tq_linear_fwd(_p(x), _p(w.contiguous()), _p(b), _p(out), B, E, N, _stream(x.device)), "linear")
    return out


# ----------------------------------------------------------------------------------------------- backward wrappers
from ._lib import TQ_BWD_ACCUM, TQ_BWD_DROPOUT, TQ_BWD_GN, TQ_BWD_SILU, TQ_BWD_STATS, TqConvBwdDesc


def conv1d_bwd_data(dy, weight, *, x0=None, x1=None, gscale=None, gshift=None, silu=False, stats=False, split=None,
                    accumulate_into=None, dropout_p=0.0, dropout_seed=0, dropout_site=0, wfmt=0, dy_amax=None):
    """dy (B,T,C_out); weight (C_out, C_in, K) torch layout.  Returns (g0, g1|None, gstats|None).
    ``wfmt``: TQ_WFMT_BF16X3 (0) or TQ_WFMT_F16_MX6 (2: dy scaled into the fp16 range by the power of two that ``dy_amax`` selects --
    an int32[1] device tensor holding the bit pattern of max|dy|; default: computed here with tq_colsum's amax output)."""
    lib = _lib.load()
    B, T, C_dy = dy.shape
    C_out, C_in, K = weight.shape
    assert C_dy == C_out
    C0 = C_in if split is None else split
    C1 = C_in - C0
    if accumulate_into is not None:
        g0, g1 = accumulate_into
    else:
        g0 = torch.empty(B, T, C0, device=dy.device)
        g1 = torch.empty(B, T, C1, device=dy.device) if C1 else None
    st = torch.empty(B, nslots(T), C_in, 2, device=dy.device) if stats else None
    d = TqConvBwdDesc()
    d.B, d.T, d.C_dy, d.C_dx0, d.C_dx1, d.ktaps = B, T, C_dy, C0, C1, K
    f = 0
    if gscale is not None:
        f |= TQ_BWD_GN
    if silu:
        f |= TQ_BWD_SILU
    if stats:
        f |= TQ_BWD_STATS
    if accumulate_into is not None:
        f |= TQ_BWD_ACCUM
    if dropout_p > 0:
        f |= TQ_BWD_DROPOUT
    d.flags = f
    d.dropout_site, d.dropout_p, d.dropout_seed = dropout_site, dropout_p, dropout_seed
    d.wfmt = wfmt
    if wfmt == _lib.TQ_WFMT_F16_MX6:
        if dy_amax is None:
            dy_amax = amax_bits(dy)
        d.dy_amax = dy_amax.data_ptr()
    wp = pack_conv_weight(weight, _lib.PACK_MODE_T[wfmt])
    check(lib.tq_conv1d_bwd_data(C.byref(d), _p(dy), _p(wp), _p(x0), _p(x1), _p(gscale), _p(gshift), _p(g0), _p(g1), _p(st),
                                 _stream(dy.device)), "conv1d_bwd_data")
    return g0, g1, st


def conv1d_bwd_weight(dy, x0, wshape, *, x1=None, gscale=None, gshift=None, silu=False, stride=1, upsample=False,
                      dropout_p=0.0, dropout_seed=0, dropout_site=0, colsum=None):
    """weight gradient; ``colsum`` = (per-sample sums (B, C_out) | None, per-channel sums (C_out,) | None): column sums of dy ACCUMULATED
    into those tensors by the same launch (tq_conv1d_bwd_weight_colsum)"""
    lib = _lib.load()
    B, T_in, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[2]
    C_out, C_in, K = wshape
    T_out = dy.shape[1]
    d = TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T_in, T_out, C0, C1, C_out
    d.ktaps, d.stride, d.pad, d.upsample = K, stride, K // 2, int(upsample)
    f = 0
    if gscale is not None:
        f |= TQ_CONV_GN
    if silu:
        f |= TQ_CONV_SILU
    if dropout_p > 0:
        f |= TQ_CONV_DROPOUT
    d.flags = f
    d.dropout_site, d.dropout_p, d.dropout_seed = dropout_site, dropout_p, dropout_seed
    ws = torch.empty(lib.tq_conv1d_bwd_weight_workspace(C.byref(d)), dtype=torch.uint8, device=dy.device)
    dw = torch.empty(C_out, C_in, K, device=dy.device)
    if colsum is not None:
        bc, c1 = colsum
        check(lib.tq_conv1d_bwd_weight_colsum(C.byref(d), _p(dy), _p(x0), _p(x1), _p(gscale), _p(gshift), _p(dw), _p(ws), ws.numel(),
                                              _p(bc), C_out if bc is not None else 0, _p(c1), None, _stream(dy.device)),
              "conv1d_bwd_weight_colsum")
        return dw
    check(lib.tq_conv1d_bwd_weight(C.byref(d), _p(dy), _p(x0), _p(x1), _p(gscale), _p(gshift), _p(dw), _p(ws), ws.numel(),
                                   _stream(dy.device)), "conv1d_bwd_weight")
    return dw


def gn_bwd_finalize(gstats, mean_rstd, gamma, T):
    lib = _lib.load()
    B, _, Cn, _ = gstats.shape
    dev = gamma.device
    a, b, c = (torch.empty(B, Cn, device=dev) for _ in range(3))
    dg, db = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    check(lib.tq_gn_bwd_finalize(_p(gstats), _p(mean_rstd), _p(gamma), B, Cn, T, _p(a), _p(b), _p(c), _p(dg), _p(db),
                                 _stream(dev)), "gn_bwd_finalize")
    return a, b, c, dg, db


def gn_bwd_apply(g, x, coefs, c_total, c_offset=0, r=None, accumulate_into=None):
    lib = _lib.load()
    B, T, Cs = g.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty_like(g)
    check(lib.tq_gn_bwd_apply(_p(g), _p(x), _p(r), _p(coefs[0]), _p(coefs[1]), _p(coefs[2]), _p(dx), B, T, Cs, c_total, c_offset,
                              int(accumulate_into is not None), _stream(g.device)), "gn_bwd_apply")
    return dx


def colsum(dy, per_sample=True, total=True, bscale=None, amax=None):
    """``amax``: optional zeroed int32[TQ_AMAX_WORDS] device block that receives the bit pattern of max|dy| (see amax_bits)"""
    lib = _lib.load()
    B, T, Cn = dy.shape
    obc = torch.zeros(B, Cn, device=dy.device) if per_sample else None
    oc = torch.zeros(Cn, device=dy.device) if total else None
    check(lib.tq_colsum(_p(dy), B, T, Cn, _p(obc), Cn, _p(oc), None, _p(bscale), _p(amax), _stream(dy.device)), "colsum")
    return obc, oc


def amax_bits(dy):
    """int32[TQ_AMAX_WORDS] device block whose maximum is the IEEE bit pattern of max|dy| (what TqConvBwdDesc.dy_amax points at), from
    tq_colsum; ``amax_value`` decodes it"""
    lib = _lib.load()
    B, T, Cn = dy.shape
    out = torch.zeros(_lib.TQ_AMAX_WORDS, dtype=torch.int32, device=dy.device)
    check(lib.tq_colsum(_p(dy), B, T, Cn, None, 0, None, None, None, _p(out), _stream(dy.device)), "colsum (amax)")
    return out


def amax_value(block) -> float:
    """max|.| recorded in a TQ_AMAX_WORDS block (host float; synchronises)"""
    return float(block.max().view(torch.float32).item()) if block.numel() > 1 else float(block.view(torch.float32).item())


def gn_bwd_apply_colsum(g, x, coefs, c_total, c_offset=0, r=None, accumulate_into=None, per_sample=True, total=True, amax=None):
    """tq_gn_bwd_apply_colsum: dx as gn_bwd_apply, plus (per-sample column sums | None, total column sums | None) of dx"""
    lib = _lib.load()
    B, T, Cs = g.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty_like(g)
    obc = torch.zeros(B, Cs, device=g.device) if per_sample else None
    oc = torch.zeros(Cs, device=g.device) if total else None
    check(lib.tq_gn_bwd_apply_colsum(_p(g), _p(x), _p(r), _p(coefs[0]), _p(coefs[1]), _p(coefs[2]), _p(dx), B, T, Cs, c_total, c_offset,
                                     int(accumulate_into is not None), _p(obc), Cs, _p(oc), None, _p(amax), _stream(g.device)),
          "gn_bwd_apply_colsum")
    return dx, obc, oc


def zero_stuff(dy, T_in):
    lib = _lib.load()
    B, T_out, Cn = dy.shape
    out = torch.empty(B, T_in, Cn, device=dy.device)
    check(lib.tq_zero_stuff(_p(dy), _p(out), B, T_out, T_in, Cn, _stream(dy.device)), "zero_stuff")
    return out


def pair_sum(d_up, accumulate_into=None):
    lib = _lib.load()
    B, T2, Cn = d_up.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty(B, T2 // 2, Cn, device=d_up.device)
    check(lib.tq_pair_sum(_p(d_up), _p(dx), B, T2 // 2, Cn, int(accumulate_into is not None), _stream(d_up.device)), "pair_sum")
    return dx


def stem_conv_bwd_weight(dy, x_nct, wshape, in_scale=None, workspace=True):
    """``workspace``: two-stage sums through a scratch buffer (default) instead of atomics onto the 30 cache lines of dw"""
    lib = _lib.load()
    B, Cin, T = x_nct.shape
    Cout, _, K = wshape
    dw = torch.zeros(Cout, Cin, K, device=dy.device)
    ws = torch.empty(lib.tq_stem_head_bwd_workspace(), dtype=torch.uint8, device=dy.device) if workspace else None
    check(lib.tq_stem_conv_bwd_weight_ws(_p(dy), _p(x_nct), _p(in_scale), _p(dw), B, Cin, T, Cout, K, _p(ws), 0 if ws is None else ws.numel(),
                                         _stream(dy.device)), "stem wgrad")
    return dw


def head_conv_bwd(dpred_nct, x, weight, gscale=None, gshift=None, c_out=None, stats=True, workspace=True):
    lib = _lib.load()
    B, T, Cin = x.shape
    Cout, _, K = weight.shape
    g = torch.empty_like(x)
    st = torch.empty(B, nslots(T), Cin, 2, device=x.device) if stats else None
    dw, db = torch.zeros_like(weight), torch.zeros(Cout, device=x.device)
    ws = torch.empty(lib.tq_stem_head_bwd_workspace(), dtype=torch.uint8, device=x.device) if workspace else None
    check(lib.tq_head_conv_bwd_ws(_p(dpred_nct), _p(c_out), _p(x), _p(gscale), _p(gshift), _p(weight.contiguous()), _p(g), _p(st),
                                  _p(dw), _p(db), B, T, Cin, Cout, K, _p(ws), 0 if ws is None else ws.numel(), _stream(x.device)), "head bwd")
    return g, st, dw, db
