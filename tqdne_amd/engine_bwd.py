"""Backward plan of a UNetEngine: the reverse sweep over the forward tape, pre-bound to the HIP gradient kernels.

Per block (reverse of the fusion map in engine.py):
  ResBlock        wgrad(conv2) | dgrad(conv2)[x dropout x SiLU', GN sums] | gn_bwd_finalize | gn_bwd_apply -> d h1
                  colsum(d h1) -> time-embedding + bias grads | wgrad(conv1) | dgrad(conv1)[x SiLU', GN sums, split over the
                  concat sources] | gn_bwd_finalize | skip path (identity: residual term; 1x1: wgrad + dgrad) | gn_bwd_apply
  AttentionBlock  wgrad/dgrad(proj) | attention_bwd (dq pass, dk/dv pass) | wgrad(qkv) | dgrad(qkv)[GN sums] |
                  gn_bwd_finalize | gn_bwd_apply (+ residual)
  Downsample      wgrad(stride 2) | zero_stuff + stride-1 dgrad      Upsample   wgrad(upsampled gather) | dgrad + pair_sum
  head / stem     dedicated small kernels
The tiny (B x 256) embedding-MLP backward is four launches of a job-table fp32 GEMM kernel (tq_gemm_f32_jobs, _embedding_jobs).
Tensors consumed twice (the UNet skip stack) get their gradient written by the first consumer met in the reverse
sweep and accumulated by the second (accumulate flags are resolved when the plan is built).
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List

import torch

from . import _lib
from ._lib import (PACK_MODE_T, TQ_AMAX_WORDS, TQ_BWD_ACCUM, TQ_BWD_DROPOUT, TQ_BWD_GN, TQ_BWD_SILU, TQ_BWD_STATS, TQ_WFMT_BF16X3, TQ_WFMT_F16_MX6, STAT_SLOT,
                   TqConvBwdDesc, check)


TAIL_WORDS = 2   # floats reserved behind the parameter gradients in a backward plan's flat buffer (see BackwardPlan.run, tail_fill)


def _p(t):
    return None if t is None else t.data_ptr()


# Column sums of dy (bias / time-embedding gradients) inside the weight-gradient launch (tq_conv1d_bwd_weight_colsum) instead of a pass
# of their own: built and parity-tested, but measured SLOWER on the paper UNet (train step 32.1 vs 31.0 ms, same box): the LDS
# adds and the extra barrier cost the k = 5 kernel more than the 1.9 ms of tq_colsum launches they replace.  Default off.
FUSE_COLSUM = __import__("os").environ.get("TQDNE_FUSE_COLSUM", "0") != "0"


# Weight-gradient launches on a second HIP stream next to the rest of the sweep (see BackwardPlan.run): measured -0.9 ms on the
# 27 ms training step of the paper UNet at B = 64 (same box, twice).  TQDNE_BWD_STREAMS=1: everything on one stream.
BWD_STREAMS = int(__import__("os").environ.get("TQDNE_BWD_STREAMS", "2"))
BWD_HIPRIO = __import__("os").environ.get("TQDNE_BWD_HIPRIO", "0") == "1"

# Round 4.  (1) Column sums (and max|.|) of a gradient tensor inside the tq_gn_bwd_apply launch that writes it last, instead of a
# tq_colsum pass of their own (TQDNE_FUSE_APPLY_COLSUM=0: separate passes).  (2) Data gradients in the fp16 + MX-fp6 scheme on dy
# scaled by a power of two taken from that max (TqConvBwdDesc.wfmt / dy_amax), where the shape allows (64 | C_dy, 128 | C_dx);
# TQDNE_DGRAD_SCHEME=bf16x3 keeps round 3's three-product scheme everywhere.
FUSE_APPLY_COLSUM = __import__("os").environ.get("TQDNE_FUSE_APPLY_COLSUM", "1") != "0"
DGRAD_SCHEME = __import__("os").environ.get("TQDNE_DGRAD_SCHEME", "f16mx6").lower()
DGRAD_MX6_C64 = __import__("os").environ.get("TQDNE_DGRAD_MX6_C64", "0") == "1"   # (off by default: see _lib.MX6_C64)
N_AMAX = 160   # blocks for max|dy| (one per gradient tensor that feeds a data gradient), kept in the tail of the flat buffer


def _nslots(T):
    return (T + STAT_SLOT - 1) // STAT_SLOT


class BackwardPlan:
    def __init__(self, eng):
        self.e = eng
        self.lib = eng.lib
        self.B, self.dev = eng.B, eng.dev
        self.m = eng.m
        self._keep = []
        self.ops: List = []
        self.bwd_dropout_descs: List[TqConvBwdDesc] = []
        self._scratch: Dict = {}
        self.op_flops: Dict[int, int] = {}   # index into self.ops -> algorithmic FLOP of that launch (bench tables)
        self._trace = None                   # list: HIP-event pairs around every launch of the next sweeps (measurement only)
        self._layout_gradients()
        self._build()
        eng.enable_block_kv()   # (from the next forward on; this plan's first sweep still re-derives the planes)

    # ------------------------------------------------------------------ memory
    def _empty(self, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.dev)
        self._keep.append(t)
        return t

    def scratch(self, tag, *shape):
        key = (tag,) + tuple(shape)
        t = self._scratch.get(key)
        if t is None:
            t = self._empty(self.B, *shape)
            self._scratch[key] = t
        return t

    def grad(self, act):
        if act.grad is None:
            act.grad = self._empty(self.B, act.T, act.C)
        return act.grad

    def _readiness_keys(self):
        """id(param) -> position of the reverse sweep at which its gradient becomes final: 0 = head / output layer, then the
        tape entries from the last block back to the first, then the stem, and last whatever the sweep does not finalise
        itself (the embedding MLPs and the ResBlocks' embedding projections: one GEMM chain after the sweep)."""
        e, m = self.e, self.m
        n = len(e.tape)
        key = {}

        def mark(ps, k):
            for p in ps:
                if p is not None:
                    key.setdefault(id(p), k)

        head = [m.out] if hasattr(m, "out") else [m.output_layer]
        for mod in head:
            mark(mod.parameters(), 0)
        for k, (kind, t) in enumerate(e.tape):
            pos = n - k
            if kind == "res":
                mark([p for name, p in t["rb"].named_parameters() if not name.startswith("emb_layers")], pos)
            elif kind == "attn":
                mark(t["ab"].parameters(), pos)
            else:
                mark([t["rec"].site.weight, t["rec"].site.bias], pos)
        stem = m.input_blocks[0][0] if hasattr(m, "input_blocks") else m.input_layer
        mark(stem.parameters(), n + 1)
        return key, n + 2

    def _layout_gradients(self):
        """One flat fp32 buffer, laid out in the order the reverse sweep finalises the gradients (so that contiguous buckets
        can be handed to the gradient exchange while the sweep is still running, DataParallelTrainer / SURVEY.md 8e):
        [head | last block ... first block | stem | emb_layers weights (contiguous, block order) | emb_layers biases |
         embedding MLPs | -- end of the parameter gradients, ``n_grad`` -- | d emb_all (scratch)]."""
        e, m = self.e, self.m
        named = list(m.named_parameters())
        self.param_order = [p for _, p in named]
        emb_w, emb_b = [], []
        for rb in e.res_blocks:
            if hasattr(rb, "emb_layers"):
                emb_w.append(rb.emb_layers[1].weight)
                emb_b.append(rb.emb_layers[1].bias)
        emb_ids = {id(p) for p in emb_w + emb_b}
        key, last = self._readiness_keys()
        rest = [p for _, p in named if id(p) not in emb_ids]
        order = sorted(range(len(rest)), key=lambda i: (key.get(id(rest[i]), last), i))
        swept = [rest[i] for i in order if key.get(id(rest[i]), last) < last]
        tail = [rest[i] for i in order if key.get(id(rest[i]), last) >= last]
        total, offs = 0, {}
        for p in swept:
            total = (total + 63) // 64 * 64
            offs[id(p)] = total
            total += p.numel()
        total = (total + 63) // 64 * 64
        self.off_emb = total
        for p in emb_w + emb_b:   # no padding in between: their gradient is ONE GEMM / one column sum over the block
            offs[id(p)] = total
            total += p.numel()
        for p in tail:
            total = (total + 63) // 64 * 64
            offs[id(p)] = total
            total += p.numel()
        self.n_grad = total
        total += TAIL_WORDS      # flat[n_grad : n_grad + TAIL_WORDS]: rides at the end of the LAST gradient bucket (``tail_fill`` of run())
        total = (total + 63) // 64 * 64
        self.off_demb = total
        total += self.B * e.emb_total
        total = (total + 63) // 64 * 64
        off_amax = total
        total += N_AMAX * TQ_AMAX_WORDS   # (zeroed with the buffer at the start of every sweep)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.amax = self.flat[off_amax:off_amax + N_AMAX * TQ_AMAX_WORDS].view(torch.int32)   # bit patterns of max|dy|, written by atomic max
        self._amax_slot = {}       # data_ptr of a gradient tensor -> index of its slot
        self.dgrad_descs = []      # every data-gradient descriptor of the plan (introspection: tests assert their scheme)
        self._mx6_dgrads = []      # (descriptor, site) of the data gradients planned in the fp16 + MX-fp6 scheme
        self.gview = {id(p): self.flat[offs[id(p)]:offs[id(p)] + p.numel()].view_as(p) for p in swept + emb_w + emb_b + tail}
        self.demb_all = self.flat[self.off_demb:self.off_demb + self.B * e.emb_total].view(self.B, e.emb_total)
        o = self.off_emb
        self.g_emb_w = self.flat[o:o + e.emb_total * e.E].view(e.emb_total, e.E)
        self.g_emb_b = self.flat[o + e.emb_total * e.E:o + e.emb_total * e.E + e.emb_total]
        self.offs = offs
        self._ready = {}        # id(param) -> index of the op of self.ops that finalises its gradient (recorded by g())
        self._swept_ids = {id(p) for p in swept}

    # ------------------------------------------------------------------ gradient buckets (overlap with the exchange)
    END = 1 << 30

    def plan_buckets(self, bucket_elems: int):
        """Cut [0, n_grad) of the flat buffer into contiguous buckets of >= ``bucket_elems`` floats and find, for each, the op
        of the sweep after which all of it is final.  Returns [(lo, hi, op_index)]; op_index END = only after the sweep
        (stem weight gradient, embedding backward)."""
        spans = []
        for p in self.param_order:
            o = self.offs[id(p)]
            r = self._ready.get(id(p), self.END) if id(p) in self._swept_ids else self.END
            if r >= len(self.ops):
                r = self.END
            if not p.requires_grad:
                r = -1
            spans.append((o, o + p.numel(), r))
        spans.sort()
        buckets, lo, ready = [], 0, -1
        for o, hi, r in spans:
            ready = max(ready, r)
            if hi - lo >= bucket_elems:
                hi_al = min((hi + 63) // 64 * 64, self.n_grad)
                buckets.append((lo, hi_al, ready))
                lo = hi_al
        if lo < self.n_grad:
            buckets.append((lo, self.n_grad, max(ready, spans[-1][2]) if buckets else ready))
        # a bucket can never be released before an earlier one (the exchange is issued in one order on every rank)
        out, run_max = [], -1
        for lo_, hi_, r in buckets:
            run_max = max(run_max, r)
            out.append((lo_, hi_, run_max))
        return out

    def amax_ptr(self, dy_ptr: int, create: bool = True):
        """device address of the slot holding max|dy| of the gradient tensor at ``dy_ptr`` (None if it has none and not ``create``)"""
        i = self._amax_slot.get(dy_ptr)
        if i is None:
            if not create or len(self._amax_slot) >= N_AMAX:
                return None
            i = self._amax_slot[dy_ptr] = len(self._amax_slot)
        return self.amax.data_ptr() + 4 * TQ_AMAX_WORDS * i

    def g(self, param):
        """the gradient view of ``param``; called while an op is being assembled, so it also records that op (the next one
        appended to self.ops) as the last writer of that gradient"""
        self._ready[id(param)] = len(self.ops)
        return self.gview[id(param)]

    def gv(self, param):
        return self.gview[id(param)]

    # ------------------------------------------------------------------ emitters
    def _wgrad(self, rec, dy, bias_colsum=True, colsum=None):
        """weight gradient of one conv; the column sums of dy that the block needs anyway (bias gradients, the per-sample gradient of
        the broadcast time embedding) ride in the same launch: ``colsum`` = (per-sample destination address | None, its row stride,
        bias gradient tensor | None, second bias gradient tensor | None); ``bias_colsum``: default = this conv's own bias."""
        lib, site = self.lib, rec.site
        need = lib.tq_conv1d_bwd_weight_workspace(C.byref(rec.desc))
        self.ws_bytes = max(getattr(self, "ws_bytes", 0), need)
        self._wgrad_ops.append(len(self.ops))
        s0 = rec.srcs[0]
        s1 = rec.srcs[1] if len(rec.srcs) > 1 else None
        if colsum is None and bias_colsum and site.bias is not None:
            colsum = (None, 0, site.bias, None)
        bc, stride, c1, c2 = colsum if colsum is not None else (None, 0, None, None)
        if FUSE_COLSUM is False and colsum is not None:   # A/B switch: the column sums as their own pass over dy
            self._colsum_pass(dy, rec.out.T if rec.out is not None else rec.desc.T_out, site.C_out, bc, stride, c1, c2, site.name)
            bc, stride, c1, c2 = None, 0, None, None
            self._wgrad_ops[-1] = len(self.ops)
        self.op_flops[len(self.ops)] = 2 * site.C_in * site.C_out * site.K * rec.desc.T_out * self.B
        self.ops.append([lib.tq_conv1d_bwd_weight_colsum, [C.byref(rec.desc), _p(dy), _p(s0.buf), _p(s1.buf) if s1 else None,
                                                           _p(rec.gn[0]) if rec.gn else None, _p(rec.gn[1]) if rec.gn else None,
                                                           _p(self.g(site.weight)), None, 0, bc, stride,
                                                           _p(self.g(c1)) if c1 is not None else None,
                                                           _p(self.g(c2)) if c2 is not None else None], "wgrad:" + site.name])

    def _colsum_pass(self, dy, T_dy, C_dy, bc, stride, c1, c2, name):
        """column sums (bias / per-sample embedding gradients) and max|dy| of the gradient tensor ``dy`` (B, T_dy, C_dy): inside the
        tq_gn_bwd_apply launch that writes dy last where there is one, else a tq_colsum pass of their own"""
        lib = self.lib
        amax = self.amax_ptr(dy.data_ptr())
        wr = self._grad_writer.get(dy.data_ptr())
        if FUSE_APPLY_COLSUM and wr is not None and wr[0][0] is lib.tq_gn_bwd_apply:
            # the launch that writes dy last is a tq_gn_bwd_apply: it forms the sums from its registers (its op is rewritten in
            # place; the gradients are final no earlier than before, so g() still records the current position)
            op = wr[0]
            op[0] = lib.tq_gn_bwd_apply_colsum
            op[1] = op[1] + [bc, stride, _p(self.g(c1)) if c1 is not None else None, _p(self.g(c2)) if c2 is not None else None, amax]
            op[2] = "gn_bwd_apply+colsum:" + name
        else:
            self.ops.append([lib.tq_colsum, [_p(dy), self.B, T_dy, C_dy, bc, stride,
                                             _p(self.g(c1)) if c1 is not None else None, _p(self.g(c2)) if c2 is not None else None, None, amax],
                             "colsum:" + name])

    def _dgrad(self, rec, dy, T, dsts, accumulate, chain=True, stats=True, amax_of=None):
        """dy (B,T,C_out of the forward conv) -> gradient wrt the forward conv's (activated) inputs.
        chain=True applies the forward prologue's derivative and emits GN sums; returns the gstats buffer (or None).
        ``amax_of``: the tensor whose max|.| slot bounds dy (default dy itself; the zero-stuffed copy of a Downsample's gradient
        has the maximum of the tensor it was made from)."""
        lib, site = self.lib, rec.site
        # scheme: fp16 + MX-fp6 on the scaled dy where the kernel is built for the shape and a max|dy| slot exists (filled by the
        # column-sum pass that every conv with a bias has ahead of its data gradient)
        amax = self.amax_ptr((amax_of if amax_of is not None else dy).data_ptr(), create=False)
        # (and only where the forward scheme requested for the model is the same one: TQDNE_CONV_SCHEME=bf16x3 / f16mx8 means
        # fp32-range three-product data gradients too)
        # (round 6: 64 | C_in through the 64-channel tile -- the 64- and 192-channel inputs of the T = 4096 level -- with TQDNE_DGRAD_MX6_C64=1; measured slower)
        cin_ok = site.C_in % 128 == 0 or (site.C_in % 64 == 0 and DGRAD_MX6_C64)
        mx6 = (DGRAD_SCHEME == "f16mx6" and _lib.requested_scheme() == "f16mx6"
               and amax is not None and site.C_out % 64 == 0 and cin_ok
               and getattr(self.e, "scheme", "auto") == "auto" and getattr(self.m, "_conv_scheme", "auto") == "auto")
        want = 5 if mx6 else 1
        if site.packed_t is None:
            nb = max(lib.tq_conv_weight_pack_bytes(site.C_out, site.C_in, site.K, 1),
                     lib.tq_conv_weight_pack_bytes(site.C_out, site.C_in, site.K, 5) if site.C_out % 64 == 0 else 0)
            site.packed_t = torch.empty(nb, dtype=torch.uint8, device=self.dev)
            site.pack_mode_t = want
            self.e.dgrad_sites.append(site)
        elif site.pack_mode_t != want:
            raise RuntimeError(f"{site.name}: data gradient planned twice with different weight formats")
        d = TqConvBwdDesc()
        d.wfmt = TQ_WFMT_F16_MX6 if mx6 else TQ_WFMT_BF16X3
        d.dy_amax = amax if mx6 else None
        self.dgrad_descs.append(d)
        if mx6:
            self._mx6_dgrads.append((d, site))
        d.B, d.T, d.C_dy = self.B, T, site.C_out
        d.C_dx0 = dsts[0].shape[2]
        d.C_dx1 = dsts[1].shape[2] if len(dsts) > 1 else 0
        assert d.C_dx0 + d.C_dx1 == site.C_in
        d.ktaps = site.K
        f = TQ_BWD_ACCUM if accumulate else 0
        gst = None
        s0 = rec.srcs[0]
        s1 = rec.srcs[1] if len(rec.srcs) > 1 else None
        if chain:
            if rec.gn is not None:
                f |= TQ_BWD_GN
                if stats:
                    f |= TQ_BWD_STATS
                    gst = self._empty(self.B, _nslots(T), site.C_in, 2)
            if rec.silu:
                f |= TQ_BWD_SILU
            if rec.dropout:
                self.bwd_dropout_descs.append((d, rec.desc))
        d.flags = f
        d.dropout_site = rec.desc.dropout_site
        self._keep.append(d)
        self.op_flops[len(self.ops)] = 2 * site.C_in * site.C_out * site.K * (rec.desc.T_out if rec.stride == 2 else T) * self.B
        self.ops.append([lib.tq_conv1d_bwd_data, [C.byref(d), _p(dy), _p(site.packed_t), _p(s0.buf) if chain else None,
                                                  _p(s1.buf) if (chain and s1) else None,
                                                  _p(rec.gn[0]) if (chain and rec.gn) else None,
                                                  _p(rec.gn[1]) if (chain and rec.gn) else None, _p(dsts[0]),
                                                  _p(dsts[1]) if len(dsts) > 1 else None, _p(gst)], "dgrad:" + site.name])
        for t_ in dsts:
            self._wrote(t_)
        return gst

    def _gn_bwd(self, gst, gn, norm, T, Ctot):
        """finalise -> coefficient arrays; accumulates dgamma/dbeta."""
        ca, cb, cc = self._empty(self.B, Ctot), self._empty(self.B, Ctot), self._empty(self.B, Ctot)
        self.ops.append([self.lib.tq_gn_bwd_finalize, [_p(gst), _p(gn[2]), _p(norm.weight), self.B, Ctot, T, _p(ca), _p(cb), _p(cc),
                                                       _p(self.g(norm.weight)), _p(self.g(norm.bias))], "gn_bwd_finalize"])
        return ca, cb, cc

    def _gn_apply(self, G, act, coef, Ctot, coff, r=None):
        dx = self.grad(act)
        self.ops.append([self.lib.tq_gn_bwd_apply, [_p(G), _p(act.buf), _p(r), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(dx), self.B,
                                                    act.T, act.C, Ctot, coff, int(act.gw)], "gn_bwd_apply"])
        self._wrote(dx)
        act.gw = True

    def _wrote(self, grad_tensor):
        """the op just appended is (so far) the last writer of ``grad_tensor``: a later consumer that needs the tensor's column
        sums may fold them into it (see _wgrad)"""
        self._grad_writer[grad_tensor.data_ptr()] = (self.ops[-1],)

    # ------------------------------------------------------------------ plan
    def _build(self):
        e, m, lib, B = self.e, self.m, self.lib, self.B
        self._wgrad_ops = []
        self._grad_writer = {}   # data_ptr of a gradient tensor -> (op entry of its last writer so far,)
        if not hasattr(e, "dgrad_sites"):
            e.dgrad_sites = []
        for a in e.acts:
            a.gw = False
        # ---- head
        final = e.final
        head = m.out[2]
        Gh = self.scratch("G", final.T, final.C)
        gst = self._empty(B, _nslots(final.T), final.C, 2)
        # (scratch for the two-stage sums of the head / stem weight gradients: one buffer each, see tq_stem_head_bwd_workspace)
        nws = lib.tq_stem_head_bwd_workspace()
        self._ws_head = torch.empty(nws, dtype=torch.uint8, device=self.dev)
        self._ws_stem = torch.empty(nws, dtype=torch.uint8, device=self.dev)
        self.head_op = [lib.tq_head_conv_bwd_ws, [None, None, _p(final.buf), _p(e.head_gn[0]), _p(e.head_gn[1]), _p(head.weight), _p(Gh),
                                                  _p(gst), _p(self.g(head.weight)), _p(self.g(head.bias)), B, final.T, final.C,
                                                  m.out_channels, head.kernel_size[0], _p(self._ws_head), nws], "head bwd"]
        coef = self._gn_bwd(gst, e.head_gn, m.out[0], final.T, final.C)
        self._gn_apply(Gh, final, coef, final.C, 0)
        # ---- blocks, reversed
        for kind, t in reversed(e.tape):
            getattr(self, "_bwd_" + kind)(t)
        # ---- stem (dynamic: x, in_scale)
        stem = m.input_blocks[0][0]
        so = e.stem_out
        assert so.gw
        self.stem_op = [lib.tq_stem_conv_bwd_weight_ws, [_p(so.grad), None, None, _p(self.g(stem.weight)), B, m.in_channels, so.T,
                                                         stem.out_channels, stem.kernel_size[0], _p(self._ws_stem), nws], "stem wgrad"]
        self._ready[id(stem.weight)] = self.END  # (run after the sweep, not from self.ops)
        # wide stems (the latent UNet's 16 input channels: 64 x 16 x 5 weights exceed the dedicated kernel's register budget)
        # are differentiated as a generic fused conv over a (B, T, 32) channels-last copy of the pre-scaled input
        self.stem_generic = stem.out_channels * m.in_channels * stem.kernel_size[0] > 2048
        if self.stem_generic:
            from ._lib import TqConvDesc
            K = stem.kernel_size[0]
            self.stem_x_btc = self._empty(B, so.T, 32)
            self.stem_x_btc.zero_()
            d = TqConvDesc()
            d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, so.T, so.T, 32, 0, stem.out_channels
            d.ktaps, d.stride, d.pad, d.upsample, d.flags = K, 1, K // 2, 0, 0
            self._keep.append(d)
            self.dw_stem32 = self._empty(stem.out_channels, 32, K)
            self.ws_bytes = max(getattr(self, "ws_bytes", 0), lib.tq_conv1d_bwd_weight_workspace(C.byref(d)))
            self._wgrad_ops.append(len(self.ops))
            self.ops.append([lib.tq_conv1d_bwd_weight, [C.byref(d), _p(so.grad), _p(self.stem_x_btc), None, None, None,
                                                        _p(self.dw_stem32), None, 0], "wgrad:stem (generic)"])
        self.ops.append([lib.tq_colsum, [_p(so.grad), B, so.T, so.C, None, 0, _p(self.g(stem.bias)), None, None, None], "colsum:stem"])
        # shared workspace of the weight-gradient slabs
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=self.dev)
        for i in self._wgrad_ops:
            self.ops[i][1][7] = self.ws.data_ptr()
            self.ops[i][1][8] = self.ws.numel()

    def _recompute(self, t):
        """use_checkpoint plans: re-issue the block's forward launches (same pre-bound calls as the forward plan's) so that its shared
        intermediate buffers hold THIS block's activations again.  The run loop makes the main stream wait for the weight-gradient
        stream in front of them: a weight gradient of the block swept before may still be reading the shared buffers."""
        for i in t.get("recompute") or ():
            fn, args, what, _fl = self.e.ops[i]
            self.ops.append([fn, list(args), "recompute:" + what])

    def _bwd_res(self, t):
        self._recompute(t)
        rb, srcs, h1, out = t["rb"], t["srcs"], t["h1"], t["out"]
        rec1, rec2, rec_sk = t["rec1"], t["rec2"], t["rec_sk"]
        B, T, Co = self.B, out.T, out.C
        assert out.gw, "gradient of a block output must be complete before its backward"
        dout = out.grad
        # conv2 (out_layers.3): weight grad, then data grad chained through dropout / SiLU / GN2.  The bias gradients of conv2
        # and of the 1x1 skip conv are the same column sums of d out: one pass
        self._wgrad(rec2, dout, colsum=(None, 0, rec2.site.bias, rec_sk.site.bias if rec_sk is not None else None))
        G2 = self.scratch("G", T, Co)
        gst2 = self._dgrad(rec2, dout, T, [G2], accumulate=False)
        coef2 = self._gn_bwd(gst2, t["g2"], rb.out_layers[0], T, Co)
        h1.gw = False
        self._gn_apply(G2, h1, coef2, Co, 0)  # d h1  (= gradient of conv1's output and of the broadcast embedding)
        dh1 = h1.grad
        # column sums of d h1: per-sample -> gradient of the broadcast time embedding; total -> bias of conv1
        emb_dst = None
        if hasattr(rb, "emb_layers"):
            emb_dst = self.demb_all.data_ptr() + 4 * self.e.emb_offsets[id(rb)]
        # conv1 (in_layers.2)
        self._wgrad(rec1, dh1, colsum=(emb_dst, self.e.emb_total, rec1.site.bias, None))
        Ctot = sum(s.C for s in srcs)
        G1 = [self.scratch("G1_%d" % i, T, s.C) for i, s in enumerate(srcs)]
        gst1 = self._dgrad(rec1, dh1, T, G1, accumulate=False)
        coef1 = self._gn_bwd(gst1, t["g1"], rb.in_layers[0], T, Ctot)
        # skip path + GN1 path into the block inputs
        if rec_sk is None:
            self._gn_apply(G1[0], srcs[0], coef1, Ctot, 0, r=dout)
        else:
            self._wgrad(rec_sk, dout, bias_colsum=False)
            acc = srcs[0].gw
            assert all(s.gw == acc for s in srcs)
            self._dgrad(rec_sk, dout, T, [self.grad(s) for s in srcs], accumulate=acc, chain=False)
            for s in srcs:
                s.gw = True
            coff = 0
            for Gs, s in zip(G1, srcs):
                self._gn_apply(Gs, s, coef1, Ctot, coff)
                coff += s.C

    def _bwd_attn(self, t):
        self._recompute(t)
        ab, x, qkv, att, out = t["ab"], t["x"], t["qkv"], t["att"], t["out"]
        B, T, Cc = self.B, x.T, x.C
        assert out.gw
        dout = out.grad
        self._wgrad(t["rec_proj"], dout)
        datt = self.grad(att)
        self._dgrad(t["rec_proj"], dout, T, [datt], accumulate=False, chain=False)
        dqkv = self.grad(qkv)
        delta = self._empty(B, ab.num_heads, T)
        self.op_flops[len(self.ops)] = 2 * 4 * ab.channels * T * T * B
        # second-generation kernels (D = 32 / 64) take a scratch buffer for the bf16 planes of Q, K, V, dO: one per shape, shared by
        # the blocks of the sweep (2 * tq_attention_workspace_bytes = 16 H Tp D bytes per sample)
        Tp = (T + 63) // 64 * 64
        # (D = 128 falls through to the first-generation kernels, which never touch it: no buffer, NULL)
        ws = self.scratch("attn_bwd_ws", 4 * ab.num_heads * Tp * t["D"]) if t["D"] in (32, 64) else None
        if ws is not None:
            # the K / V planes of the training forward are re-used where that forward kept them (engine.enable_block_kv: every forward
            # after this plan was built); ``t["kv_ws"]`` is filled in by then
            e, lib = self.e, self.lib

            def attn_bwd(qkv_p, att_p, datt_p, lse_p, delta_p, dqkv_p, ws_p, B_, T_, H_, D_, stream, _t=t):
                kv = _t.get("kv_ws") if (e._last.get("block_kv") or _t.get("kv_always")) else None
                if kv is not None:
                    return lib.tq_attention_bwd_ws_kv(qkv_p, att_p, datt_p, lse_p, delta_p, dqkv_p, ws_p, kv.data_ptr(), B_, T_, H_, D_, stream)
                return lib.tq_attention_bwd_ws(qkv_p, att_p, datt_p, lse_p, delta_p, dqkv_p, ws_p, B_, T_, H_, D_, stream)
            fn = attn_bwd
        else:
            fn = self.lib.tq_attention_bwd_ws
        self.ops.append([fn, [_p(qkv.buf), _p(att.buf), _p(datt), _p(t["lse"]), _p(delta), _p(dqkv), _p(ws),
                              B, T, ab.num_heads, t["D"]], "attention bwd"])
        self._wrote(dqkv)
        self._wgrad(t["rec_qkv"], dqkv)
        G = self.scratch("G", T, Cc)
        gst = self._dgrad(t["rec_qkv"], dqkv, T, [G], accumulate=False)
        coef = self._gn_bwd(gst, t["g"], ab.norm, T, Cc)
        self._gn_apply(G, x, coef, Cc, 0, r=dout)

    def _bwd_down(self, t):
        x, out, rec = t["x"], t["out"], t["rec"]
        assert out.gw
        dout = out.grad
        self._wgrad(rec, dout)
        dyz = self.scratch("dyz", x.T, out.C)
        self.ops.append([self.lib.tq_zero_stuff, [_p(dout), _p(dyz), self.B, out.T, x.T, out.C], "zero_stuff"])
        self._dgrad(rec, dyz, x.T, [self.grad(x)], accumulate=x.gw, chain=False, amax_of=dout)
        x.gw = True

    def _bwd_up_poly(self, t):
        """Upsample whose training forward ran in the two-phase k = 3 form (engine.POLY_TRAIN): the gradients of THAT conv, with the
        output gradient (B, 2T, C) read as (B, T, 2C) -- row m of the view = [row 2m | row 2m + 1] = the two phases' channel blocks.
          bias:    column sums of d out over the real (B, 2T, C) view (+ its max|.| for the fp16-range data gradient)
          weights: k = 3 weight gradient d W2 (2C, C_in, 3), folded onto the five taps (tq_upsample_poly_wgrad_fold)
          input:   k = 3 data gradient with the transposed two-phase weights, written (or accumulated) straight into d x"""
        from ._lib import TqConvDesc
        from .engine import ConvRec
        lib = self.lib
        x, out, rec = t["x"], t["out"], t["rec"]
        ps, d2 = rec.poly
        site = rec.site
        assert out.gw and len(rec.srcs) == 1
        dout = out.grad
        self._colsum_pass(dout, out.T, site.C_out, None, 0, site.bias, None, site.name)
        dw = TqConvDesc()
        dw.B, dw.T_in, dw.T_out, dw.C_in0, dw.C_in1, dw.C_out = self.B, x.T, x.T, x.C, 0, 2 * site.C_out
        dw.ktaps, dw.stride, dw.pad, dw.upsample, dw.flags = 3, 1, 1, 0, 0
        self._keep.append(dw)
        dW2 = self._empty(2 * site.C_out, site.C_in, 3)
        self.ws_bytes = max(getattr(self, "ws_bytes", 0), lib.tq_conv1d_bwd_weight_workspace(C.byref(dw)))
        self._wgrad_ops.append(len(self.ops))
        self.op_flops[len(self.ops)] = 2 * site.C_in * 2 * site.C_out * 3 * x.T * self.B
        self.ops.append([lib.tq_conv1d_bwd_weight_colsum, [C.byref(dw), _p(dout), _p(x.buf), None, None, None, _p(dW2), None, 0,
                                                           None, 0, None, None], "wgrad:" + site.name + "+polyphase"])
        # (named like a weight gradient: it follows it on the weight-gradient stream)
        self.ops.append([lib.tq_upsample_poly_wgrad_fold, [_p(dW2), _p(self.g(site.weight)), site.C_out, site.C_in],
                         "wgrad:fold:" + site.name])
        rec2 = ConvRec(ps, d2, rec.srcs, None, None, 1, False, False, False)
        self._dgrad(rec2, dout, x.T, [self.grad(x)], accumulate=x.gw, chain=False)
        x.gw = True

    def _bwd_up(self, t):
        if getattr(t["rec"], "poly", None) is not None:
            return self._bwd_up_poly(t)
        x, out, rec = t["x"], t["out"], t["rec"]
        assert out.gw
        dout = out.grad
        self._wgrad(rec, dout)
        dup = self.scratch("dup", out.T, x.C)
        self._dgrad(rec, dout, out.T, [dup], accumulate=False, chain=False)
        self.ops.append([self.lib.tq_pair_sum, [_p(dup), _p(self.grad(x)), self.B, x.T, x.C, int(x.gw)], "pair_sum"])
        self._wrote(x.grad)
        x.gw = True

    # ------------------------------------------------------------------ run
    def _input_gradient(self, last, stream):
        """d loss / d x (B, C_in, T) of the last forward: the stem conv's data gradient (reference: plain autograd through
        ``input_blocks[0]``, unet.py:233,389-391), as a generic transposed conv of d stem_out into a 32-channel channels-last buffer
        (the packer zero-fills the weight rows of the channels the model does not have), times the per-sample input scale where
        the stem load applied one (EDM's c_in).  Built on first use: training never asks for it."""
        e, m, lib = self.e, self.m, self.lib
        stem, so = m.input_blocks[0][0], e.stem_out
        cin, K = stem.in_channels, stem.kernel_size[0]
        if cin > 32:
            raise NotImplementedError("input gradient for more than 32 input channels")
        if getattr(self, "dx_op", None) is None:
            self.dx_btc = self._empty(self.B, so.T, 32)
            self.stem_packed_t = torch.empty(lib.tq_conv_weight_pack_bytes(stem.out_channels, cin, K, 1), dtype=torch.uint8, device=self.dev)
            bd = TqConvBwdDesc()
            bd.B, bd.T, bd.C_dy, bd.C_dx0, bd.C_dx1, bd.ktaps, bd.flags = self.B, so.T, stem.out_channels, 32, 0, K, 0
            self._keep.append(bd)
            self.dx_op = [lib.tq_conv1d_bwd_data, [C.byref(bd), _p(so.grad), _p(self.stem_packed_t), None, None, None, None,
                                                   _p(self.dx_btc), None, None], "dgrad:stem"]
        check(lib.tq_pack_conv_weight(stem.weight.data_ptr(), stem.out_channels, cin, K, 1, self.stem_packed_t.data_ptr(), stream),
              "pack^T stem")
        fn, args, what = self.dx_op
        check(fn(*args, stream), what)
        dx = self.dx_btc[:, :, :cin].permute(0, 2, 1).contiguous()
        if last["in_scale"] is not None:
            dx.mul_(last["in_scale"][:, None, None])
        return dx

    def run(self, *args, **kw):
        """The reverse sweep (``_run``).  TQDNE_BWD_HIPRIO=1 (experiment, round 6): the sweep's chain -- data gradients, GroupNorm backward,
        column sums -- runs on a HIGH-priority stream while the weight gradients stay on the normal-priority side stream, so that the
        chain's small launches are dispatched ahead of the weight gradients' pending workgroups instead of queueing behind them (under
        rocprofv3 a `gn_bwd_finalize` launch takes 33-43 us next to a weight gradient and 8 us alone)."""
        if BWD_HIPRIO and self._trace is None and BWD_STREAMS == 2 and not torch.cuda.is_current_stream_capturing():
            from .engine import hiprio_stream
            main_t = torch.cuda.current_stream(self.dev)
            hp = hiprio_stream(self.dev)
            hp.wait_stream(main_t)
            with torch.cuda.stream(hp):
                res = self._run(*args, **kw)
            main_t.wait_stream(hp)
            return res
        return self._run(*args, **kw)

    def _run(self, dpred: torch.Tensor, gloss: torch.Tensor, clone: bool = True, on_bucket=None, bucket_elems: int = 4 << 20,
             tail_fill=None, want_dx: bool = False):
        """``want_dx``: also form d loss / d x of the forward's input; left in ``self.last_dx`` (B, C_in, T).
        ``on_bucket(flat_slice)``: called from inside the sweep, right after the launch that finalises the last gradient of
        each bucket of >= ``bucket_elems`` floats has been enqueued (buckets = contiguous slices of the flat buffer in the
        order the sweep completes them; every rank cuts them identically).  The data-parallel trainer starts the slice's
        all-reduce there, so the exchange runs under the rest of the backward.
        ``tail_fill(words)``: the ``TAIL_WORDS`` floats that follow the gradients in the flat buffer are handed to this callback right
        before the LAST bucket leaves, and that bucket then includes them: a few extra words of the caller (the trainer's range-guard
        flags) ride in the gradients' last collective instead of in one of their own."""
        if tail_fill is not None and on_bucket is not None:
            user_bucket, n_grad, flat = on_bucket, self.n_grad, self.flat

            def on_bucket(sl):   # (the bucket that ends the gradients takes the tail words along)
                end = sl.storage_offset() + sl.numel()
                if end == n_grad:
                    tail_fill(flat[n_grad:n_grad + TAIL_WORDS])
                    sl = flat[sl.storage_offset():n_grad + TAIL_WORDS]
                user_bucket(sl)
        e, m, lib = self.e, self.m, self.lib
        last = e._last
        if last.get("infer", False):
            raise RuntimeError("the last forward of this plan was an inference forward (infer=True): it kept nothing for a backward")
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self._follow_scheme()
        e.repack_transposed(stream)
        self.flat.zero_()
        p, seed = float(last["dropout_p"]), int(last["dropout_seed"])
        for d, fd in self.bwd_dropout_descs:
            if p > 0.0:
                d.flags |= TQ_BWD_DROPOUT
                d.dropout_p, d.dropout_seed = p, seed
            else:
                d.flags &= ~TQ_BWD_DROPOUT
        c_out = last["c_out"]
        gl = gloss.to(torch.float32).reshape(())
        cs = (c_out * gl) if c_out is not None else gl.expand(self.B).contiguous()
        self._keep_run = (cs, dpred)
        fn, args, what = self.head_op
        args[0], args[1] = dpred.data_ptr(), cs.data_ptr()
        check(fn(*args, stream), what)
        cin = m.in_channels
        if self.stem_generic:
            xs = last["x"] if last["in_scale"] is None else last["x"] * last["in_scale"][:, None, None]
            self.stem_x_btc[:, :, :cin].copy_(xs.permute(0, 2, 1))
        if self._trace is not None:
            from .engine import _recorded_event
            # (a traced pass still hands every bucket to the gradient exchange: a trainer that traces must not step on
            # un-reduced gradients)
            fire, late = self._fire_points(bucket_elems) if on_bucket is not None else ({}, ())
            for i, (fn, args, what) in enumerate(self.ops):
                a = _recorded_event()
                rc = fn(*args, stream)
                self._trace.append((what, self.op_flops.get(i, 0), 0, a, _recorded_event()))
                if rc:
                    check(rc, what)
                if i in fire:
                    for lo, hi in fire[i]:
                        on_bucket(self.flat[lo:hi])
        elif BWD_STREAMS == 2:
            # Weight gradients on a second stream: they only read (dy, forward activations) and write their own slice of the
            # flat buffer, so the sweep's chain (data gradients, GroupNorm backward, column sums: half of it HBM-bound) does not
            # have to wait for them.  Every gradient tensor a weight-gradient launch reads is a buffer of its own, never reused.
            main_t = torch.cuda.current_stream(self.dev)
            from .engine import side_stream
            side = side_stream(self.dev, 1)   # (the sampler lanes' pool: a stream of its own would be one hardware queue too many)
            if self.__dict__.get("_side_evs") is None:
                self._side_evs = {}
            fire, late = self._fire_points(bucket_elems) if on_bucket is not None else ({}, ())
            for i, (fn, args, what) in enumerate(self.ops):
                if what.startswith("recompute:"):   # (use_checkpoint: shared block-internal buffers are about to be overwritten)
                    main_t.wait_stream(side)
                if what.startswith("wgrad:"):   # (the column sums on that stream too: measured equal)
                    ev = self._side_evs.get(i)
                    if ev is None:
                        ev = self._side_evs[i] = torch.cuda.Event()
                    ev.record(main_t)
                    side.wait_event(ev)
                    rc = fn(*args, side.cuda_stream)
                else:
                    rc = fn(*args, stream)
                if rc:
                    check(rc, what)
                if i in fire:
                    main_t.wait_stream(side)   # (the exchange waits on the main stream only)
                    for lo, hi in fire[i]:
                        on_bucket(self.flat[lo:hi])
            main_t.wait_stream(side)
        elif on_bucket is None:
            for fn, args, what in self.ops:
                rc = fn(*args, stream)
                if rc:
                    check(rc, what)
            late = ()
        else:
            fire, late = self._fire_points(bucket_elems)
            for i, (fn, args, what) in enumerate(self.ops):
                rc = fn(*args, stream)
                if rc:
                    check(rc, what)
                if i in fire:
                    for lo, hi in fire[i]:
                        on_bucket(self.flat[lo:hi])
        if self.stem_generic:
            self.gv(m.input_blocks[0][0].weight).copy_(self.dw_stem32[:, :cin, :])
        else:
            fn, args, what = self.stem_op
            args[1], args[2] = last["x"].data_ptr(), _p(last["in_scale"])
            check(fn(*args, stream), what)
        self._embedding_backward(last)
        for lo, hi in late:
            on_bucket(self.flat[lo:hi])
        self.last_dx = self._input_gradient(last, stream) if want_dx else None
        out = self.flat.clone() if clone else self.flat  # clone: autograd may keep the returned tensors alive
        res = []
        for p_ in self.param_order:
            if not p_.requires_grad:
                res.append(None)
            else:
                o = self.offs[id(p_)]
                res.append(out[o:o + p_.numel()].view_as(p_))
        return res

    def _follow_scheme(self):
        """the forward plan left the fp16-range scheme (range guard, or bf16x3 on request): the data gradients follow -- descriptors
        edited in place, the transposed fragments re-packed as bf16x3 on the next repack_transposed"""
        if self._mx6_dgrads and (self.e.scheme != "auto" or getattr(self.m, "_conv_scheme", "auto") != "auto"):
            for d, site in self._mx6_dgrads:
                d.wfmt, d.dy_amax = TQ_WFMT_BF16X3, None
                site.pack_mode_t = 1
            self._mx6_dgrads = []

    def _fire_points(self, bucket_elems):
        key = int(bucket_elems)
        cached = getattr(self, "_fire_cache", None)
        if cached is None or cached[0] != key:
            fire, late = {}, []
            for lo, hi, r in self.plan_buckets(key):
                if r >= self.END:
                    late.append((lo, hi))
                else:
                    fire.setdefault(max(r, 0), []).append((lo, hi))
            cached = (key, fire, late)
            self._fire_cache = cached
        return cached[1], cached[2]

    def _embedding_jobs(self):
        """The job tables of the embedding backward (built once: every operand is a static buffer of the plan or a parameter).
        Dependent levels, each ONE tq_gemm_f32_jobs launch:
          1  d W_proj = d emb_all^T . SiLU(emb);  d b_proj = 1^T . d emb_all;  partial products of d emb_all . W_proj
          1b d emb = (sum of the partial products) * SiLU'(emb)
          2  per MLP (time, cond): d W_2 = d emb^T . SiLU(h);  d b_2 = 1^T . d emb;  d h = (d emb . W_2) * SiLU'(h)
          3  per MLP: d W_0 = d h^T . input;  d b_0 = 1^T . d h      (input = Fourier features / cond)"""
        from ._lib import TqGemmJob
        e, m, B, dev = self.e, self.m, self.B, self.dev
        E, Et, mc = e.E, e.emb_total, m.model_channels
        self.ones = torch.ones(B, device=dev)
        self.d_emb = self._empty(B, E)
        self.dh0 = self._empty(B, E)
        self.four = self._empty(B, mc)
        tm = m.time_mlp
        h0 = e.emb_hidden[:, 0]       # (B, E) view, row stride 2E
        cm = m.cond_mlp if m.cond_features is not None else None
        if cm is not None:
            self.dc0 = self._empty(B, E)
            c0 = e.emb_hidden[:, 1]
            self.cond_buf = self._empty(B, m.cond_features)

        def job(A, sam, sak, Bm, sbk, sbn, Cm, ldc, M, N, K, U=None, ldu=0, pre_b=0):
            jb = TqGemmJob()
            jb.A, jb.B, jb.C, jb.U = A, Bm, Cm, U
            jb.M, jb.N, jb.K = M, N, K
            jb.sam, jb.sak, jb.sbk, jb.sbn, jb.ldc, jb.ldu, jb.pre_b = sam, sak, sbk, sbn, ldc, ldu, pre_b
            return jb

        demb, ones = self.demb_all, self.ones
        hid = e.emb_hidden.stride(0)
        # d emb = d emb_all . W_proj reduces over all Et = sum of the blocks' channels (5632 for the paper UNet) into only
        # B x E / 32^2 = 16 output tiles: as one job that is 16 workgroups walking 176 dependent load -> barrier -> FMA rounds
        # (measured 160 us for 0.18 GFLOP).  Split along the reduction into KS partial products (same launch as the weight /
        # bias gradients), summed and multiplied by SiLU'(emb) by a one-row "GEMM" with a vector of ones in a launch of its own.
        KS = max(1, min(int(__import__("os").environ.get("TQDNE_EMB_KSPLIT", "16")), Et // 256, B))   # (env: A/B switch, 1 = one job)
        self.d_emb_part = self._empty(KS, B, E)
        kcut = [(Et * i // KS) // 32 * 32 for i in range(KS)] + [Et]
        lv1 = [job(_p(demb), 1, Et, _p(e.silu_emb), E, 1, _p(self.g_emb_w), E, Et, E, B),          # d W_proj
               job(_p(ones), 0, 1, _p(demb), Et, 1, _p(self.g_emb_b), Et, 1, Et, B)]               # d b_proj
        for i in range(KS):
            k0, k1 = kcut[i], kcut[i + 1]
            lv1.append(job(_p(demb) + 4 * k0, Et, 1, _p(e.emb_w) + 4 * k0 * E, E, 1, _p(self.d_emb_part[i]), E, B, E, k1 - k0))
        lv1b = [job(_p(ones), 0, 1, _p(self.d_emb_part), B * E, 1, _p(self.d_emb), B * E, 1, B * E, KS, U=_p(e.emb), ldu=B * E)]
        lv2 = [job(_p(self.d_emb), 1, E, _p(h0), hid, 1, _p(self.gv(tm[2].weight)), E, E, E, B, pre_b=1),
               job(_p(ones), 0, 1, _p(self.d_emb), E, 1, _p(self.gv(tm[2].bias)), E, 1, E, B),
               job(_p(self.d_emb), E, 1, _p(tm[2].weight), E, 1, _p(self.dh0), E, B, E, E, U=_p(h0), ldu=hid)]
        lv3 = [job(_p(self.dh0), 1, E, _p(self.four), mc, 1, _p(self.gv(tm[0].weight)), mc, E, mc, B),
               job(_p(ones), 0, 1, _p(self.dh0), E, 1, _p(self.gv(tm[0].bias)), E, 1, E, B)]
        if cm is not None:
            nc = m.cond_features
            lv2 += [job(_p(self.d_emb), 1, E, _p(c0), hid, 1, _p(self.gv(cm[2].weight)), E, E, E, B, pre_b=1),
                    job(_p(ones), 0, 1, _p(self.d_emb), E, 1, _p(self.gv(cm[2].bias)), E, 1, E, B),
                    job(_p(self.d_emb), E, 1, _p(cm[2].weight), E, 1, _p(self.dc0), E, B, E, E, U=_p(c0), ldu=hid)]
            lv3 += [job(_p(self.dc0), 1, E, _p(self.cond_buf), nc, 1, _p(self.gv(cm[0].weight)), nc, E, nc, B),
                    job(_p(ones), 0, 1, _p(self.dc0), E, 1, _p(self.gv(cm[0].bias)), E, 1, E, B)]
        self._gemm_levels = []
        for jobs in (lv1, lv1b, lv2, lv3):
            total = 0
            for jb in jobs:
                jb.tile_begin = total
                total += self.lib.tq_gemm_tiles(jb.M, jb.N)
            arr = (TqGemmJob * len(jobs))(*jobs)
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._keep.append(table)
            self._gemm_levels.append((table, len(jobs), total))

    def _embedding_backward(self, last):
        """Backward of Fourier -> time MLP (+ cond MLP) -> per-block Linear(SiLU(emb)) (unet.py:91-97, 210-227, 383-388): the
        Fourier features and three tq_gemm_f32_jobs launches (see _embedding_jobs)."""
        e, m, lib = self.e, self.m, self.lib
        if getattr(self, "_gemm_levels", None) is None:
            self._embedding_jobs()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        check(lib.tq_fourier_features(_p(last["timesteps"]), _p(m.time_embed.W), _p(self.four), self.B, m.model_channels // 2, stream),
              "fourier features")
        if m.cond_features is not None:
            self.cond_buf.copy_(last["cond"])
        for table, n, total in self._gemm_levels:
            check(lib.tq_gemm_f32_jobs(table.data_ptr(), n, total, stream), "embedding backward GEMMs")


class SeqBackwardPlan(BackwardPlan):
    """Backward plan of a SeqEngine (VAE Encoder / Decoder, blocks.py:263-436; autoencoder.py:59-84): the same block
    emitters, no embedding, a plain output conv instead of the GN-SiLU head, and optionally the gradient wrt the input
    (the decoder's d latent feeds the encoder).  The input layer is differentiated as a generic fused conv over a
    channels-last copy of the input padded to 32 channels (the packer zero-fills the missing weight rows / k entries)."""

    def _build(self):
        from ._lib import TqConvDesc
        from .engine import Act, ConvRec, ConvSite
        e, m, lib, B = self.e, self.m, self.lib, self.B
        self._wgrad_ops = []
        self._grad_writer = {}   # data_ptr of a gradient tensor -> (op entry of its last writer so far,)
        if not hasattr(e, "dgrad_sites"):
            e.dgrad_sites = []
        for a in e.acts:
            a.gw = False
        final, out = e.final, m.output_layer
        dfin = self.grad(final)
        if e.out_mode == "head":   # narrow output (decoder): VALU kernel straight from the NCW gradient
            nws = lib.tq_stem_head_bwd_workspace()
            self._ws_head = torch.empty(nws, dtype=torch.uint8, device=self.dev)
            self.head_op = [lib.tq_head_conv_bwd_ws, [None, None, _p(final.buf), None, None, _p(out.weight), _p(dfin), None,
                                                      _p(self.g(out.weight)), _p(self.g(out.bias)), B, final.T, final.C,
                                                      out.out_channels, out.kernel_size[0], _p(self._ws_head), nws], "output layer bwd"]
            self.dout_btc = None
        else:                      # wide output (encoder): generic conv gradients from a channels-last copy of d out
            self.head_op = None
            self.dout_btc = self._empty(B, final.T, out.out_channels)
            self._wgrad(e.out_rec, self.dout_btc)
            self._dgrad(e.out_rec, self.dout_btc, final.T, [dfin], accumulate=False, chain=False)
        final.gw = True
        for kind, t in reversed(e.tape):
            getattr(self, "_bwd_" + kind)(t)
        # ---- input layer: generic conv over a (B, T, 32) channels-last copy of the input
        stem, so = m.input_layer, e.stem_out
        assert so.gw
        cin, K = stem.in_channels, stem.kernel_size[0]
        self.x_btc = Act(self._empty(B, so.T, 32), None, 32, so.T)
        self.x_btc.buf.zero_()
        site = ConvSite("input_layer", stem.weight, stem.bias, self.dev, lib)
        site.C_in = 32  # descriptor / packed geometry see the padded input; pack kernels guard the real (C_out, cin, K) weight
        d = TqConvDesc()
        d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, so.T, so.T, 32, 0, stem.out_channels
        d.ktaps, d.stride, d.pad, d.upsample, d.flags = K, 1, K // 2, 0, 0
        self._keep.append(d)
        rec = ConvRec(site, d, [self.x_btc], None, so, 1, False, False, False)
        self.dw_stem32 = self._empty(stem.out_channels, 32, K)
        need = lib.tq_conv1d_bwd_weight_workspace(C.byref(d))
        self.ws_bytes = max(getattr(self, "ws_bytes", 0), need)
        self._wgrad_ops.append(len(self.ops))
        self.ops.append([lib.tq_conv1d_bwd_weight, [C.byref(d), _p(so.grad), _p(self.x_btc.buf), None, None, None,
                                                    _p(self.dw_stem32), None, 0], "wgrad:input_layer"])
        self.ops.append([lib.tq_colsum, [_p(so.grad), B, so.T, so.C, None, 0, _p(self.g(stem.bias)), None, None, None], "colsum:stem"])
        # d input (only run on request): transposed conv into a 32-channel channels-last buffer
        self.dx_btc = self._empty(B, so.T, 32)
        n0 = len(self.ops)
        site.packed_t = torch.empty(lib.tq_conv_weight_pack_bytes(stem.out_channels, cin, K, 1), dtype=torch.uint8, device=self.dev)
        self.stem_site = site
        bd = TqConvBwdDesc()
        bd.B, bd.T, bd.C_dy, bd.C_dx0, bd.C_dx1, bd.ktaps, bd.flags = B, so.T, stem.out_channels, 32, 0, K, 0
        self._keep.append(bd)
        self.dx_op = [lib.tq_conv1d_bwd_data, [C.byref(bd), _p(so.grad), _p(site.packed_t), None, None, None, None,
                                               _p(self.dx_btc), None, None], "dgrad:input_layer"]
        assert len(self.ops) == n0
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=self.dev)
        for i in self._wgrad_ops:
            self.ops[i][1][7] = self.ws.data_ptr()
            self.ops[i][1][8] = self.ws.numel()

    def run_seq(self, dout: torch.Tensor, want_dx: bool = False, clone: bool = True):
        e, m, lib = self.e, self.m, self.lib
        last = e._last
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self._follow_scheme()
        e.repack_transposed(stream)
        stem = m.input_layer
        cin, K = stem.in_channels, stem.kernel_size[0]
        if want_dx:
            check(lib.tq_pack_conv_weight(stem.weight.data_ptr(), stem.out_channels, cin, K, 1, self.stem_site.packed_t.data_ptr(),
                                          stream), "pack^T input_layer")
        self.flat.zero_()
        p, seed = float(last["dropout_p"]), int(last["dropout_seed"])
        for d, fd in self.bwd_dropout_descs:
            if p > 0.0:
                d.flags |= TQ_BWD_DROPOUT
                d.dropout_p, d.dropout_seed = p, seed
            else:
                d.flags &= ~TQ_BWD_DROPOUT
        dout = dout.contiguous()
        self.x_btc.buf[:, :, :cin].copy_(last["x"].permute(0, 2, 1))
        if self.head_op is not None:
            fn, args, what = self.head_op
            args[0] = dout.data_ptr()
            check(fn(*args, stream), what)
        else:
            self.dout_btc.copy_(dout.permute(0, 2, 1))
        for fn, args, what in self.ops:
            rc = fn(*args, stream)
            if rc:
                check(rc, what)
        self.gv(stem.weight).copy_(self.dw_stem32[:, :cin, :])
        dx = None
        if want_dx:
            fn, args, what = self.dx_op
            check(fn(*args, stream), what)
            dx = self.dx_btc[:, :, :cin].permute(0, 2, 1).contiguous()
        out = self.flat.clone() if clone else self.flat
        res = []
        for p_ in self.param_order:
            if not p_.requires_grad:
                res.append(None)
            else:
                o = self.offs[id(p_)]
                res.append(out[o:o + p_.numel()].view_as(p_))
        return res, dx
