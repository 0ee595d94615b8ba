"""torch.autograd bridges: the HIP forward/backward of the EDM training step exposed as autograd Functions so
that ``loss.backward()`` / Lightning / DDP see ordinary parameter gradients.  The arithmetic is entirely in
libtqdne_hip.so; autograd only routes the resulting gradient tensors."""

from __future__ import annotations

import torch as th

from . import _lib, engine, rng
from ._lib import check


def _p(t):
    return None if t is None else t.data_ptr()


class _EDMLossFn(th.autograd.Function):
    """loss = mean(lambda(sigma) * (D(y + sigma*n; sigma) - y)^2)   (reference edm.py:126-134)."""

    @staticmethod
    def forward(ctx, module, sample, eps, unit_noise, cond, cond_sample, *params):
        lib = _lib.load()
        B = sample.shape[0]
        per = sample[0].numel()
        dev = sample.device
        stream = th.cuda.current_stream(dev).cuda_stream
        key = ("train", tuple(sample.shape), str(dev), module._lane)
        bufs = module._scal.get(key)
        if bufs is None:
            bufs = dict(sigma=th.empty(B, device=dev), x=th.empty_like(sample), loss=th.empty(1, device=dev),
                        dpred=th.empty_like(sample))
            module._scal[key] = bufs
        e = module.edm
        check(lib.tq_edm_noise_inject(_p(sample), _p(unit_noise), _p(eps), float(e.P_mean), float(e.P_std), _p(bufs["sigma"]),
                                      _p(bufs["x"]), B, per, stream), "noise inject")
        train = module.training
        seed = rng.next_dropout_seed()
        pred = module._denoise_static(bufs["x"], bufs["sigma"], 1, cond, train=train, dropout_seed=seed, cond_sample=cond_sample)
        sc = module._scalars(B, dev)
        need_grad = any(p.requires_grad for p in params)
        check(lib.tq_edm_loss(_p(pred), _p(sample), _p(sc[4]), _p(bufs["loss"]), _p(bufs["dpred"]) if need_grad else None, B,
                              per, stream), "edm loss")
        ctx.module, ctx.bufs, ctx.shape, ctx.nparams = module, bufs, tuple(sample.shape), len(params)
        ctx.cond = cond
        ctx.concat = cond_sample is not None  # the stem then saw a pre-scaled, concatenated input
        # the plan that holds this forward's activations stays with the graph (the reference keeps its activations alive the same
        # way): looked up again in backward it could be a NEW plan if the bounded cache evicted this one in between
        ctx.eng = module.unet._engine(B, sample.shape[2], dev, module._lane)
        ctx.fwd_id = ctx.eng._fwd_count
        ctx.scalars = module._scalars(B, dev)
        return bufs["loss"][0].clone()

    @staticmethod
    def backward(ctx, gloss):
        eng, bufs, sc = ctx.eng, ctx.bufs, ctx.scalars
        if eng._fwd_count != ctx.fwd_id:
            raise RuntimeError("another forward of the same shape ran between this loss and its backward: the execution plan's static "
                               "buffers no longer hold its activations (call backward before the next step of that shape)")
        grads = eng.backward(bufs["dpred"], gloss, c_out=sc[1], in_scale=None if ctx.concat else sc[0])
        return (None, None, None, None, None, None) + tuple(grads)


def edm_loss_and_grads(module, sample, eps, unit_noise, cond, cond_sample=None, lanes=None, on_bucket=None,
                       bucket_elems: int = 4 << 20, tail_fill=None):
    """Fused training step without the autograd round trip: runs the HIP forward and backward back to back and leaves the
    gradients in the backward plan's flat buffer; ``p.grad`` of every UNet parameter is (re)bound to its view of that buffer.
    Returns (loss, flat_gradient_buffer).  Used by DataParallelTrainer (one all-reduce over the flat buffer, no per-parameter
    accumulation kernels).

    ``lanes`` > 1: the batch is split into sub-batches whose forward + backward run on separate HIP streams with their own plans
    (see LightningEDM.sample_deterministically); the loss is the mean of the sub-batch losses (``gloss = 1 / lanes`` scales each
    backward) and the lanes' flat gradient buffers are summed into lane 0's.  Default ``train_lanes(B)``.

    ``on_bucket(flat_slice)``: gradient-exchange hook of the data-parallel trainer, called from inside the backward sweep as
    soon as a bucket of the flat buffer is final (BackwardPlan.run); with several lanes the buckets are only final after
    the lanes' buffers have been summed, so the hook is then called for every bucket at the end.

    ``tail_fill``: see BackwardPlan.run (extra words of the caller at the end of the last bucket)."""
    params = list(module.unet.parameters())
    B = sample.shape[0]
    if lanes is None:
        lanes = train_lanes(B)
    if lanes < 2 or B % lanes:
        lanes = 1
    dev = sample.device
    h = B // lanes
    main = th.cuda.current_stream(dev)
    streams = [main] + [module._side_stream(dev, i) for i in range(1, lanes)]
    cut = lambda t, i: None if t is None else t[i * h:(i + 1) * h].contiguous()
    losses, flats = [], []
    with th.no_grad():
        for st in streams[1:]:
            st.wait_stream(main)
        try:
            for i, st in enumerate(streams):
                module._lane = i if lanes == 1 else engine.CONCURRENT_LANE0 + i
                with th.cuda.stream(st):
                    ctx = _Ctx()
                    loss = _EDMLossFn.forward(ctx, module, cut(sample, i) if lanes > 1 else sample, cut(eps, i) if lanes > 1 else eps,
                                              cut(unit_noise, i) if lanes > 1 else unit_noise, cut(cond, i) if lanes > 1 else cond,
                                              cut(cond_sample, i) if lanes > 1 else cond_sample, *params)
                    bufs, eng = ctx.bufs, ctx.eng
                    scale = th.full((), 1.0 / lanes, device=dev)
                    grads = eng.backward(bufs["dpred"], scale, clone=False, on_bucket=on_bucket if lanes == 1 else None,
                                         bucket_elems=bucket_elems, tail_fill=tail_fill if lanes == 1 else None)
                    losses.append(loss)
                    flats.append(eng._bwd.flat)
                    if i == 0:
                        grads0, eng0_bwd = grads, eng._bwd
        finally:
            module._lane = 0
        for i, st in enumerate(streams[1:], 1):
            main.wait_stream(st)
        for f in flats[1:]:
            flats[0].add_(f)
        if on_bucket is not None and lanes > 1:
            from .engine_bwd import TAIL_WORDS
            for lo, hi, _ in eng0_bwd.plan_buckets(bucket_elems):
                if tail_fill is not None and hi == eng0_bwd.n_grad:
                    tail_fill(flats[0][hi:hi + TAIL_WORDS])
                    hi += TAIL_WORDS
                on_bucket(flats[0][lo:hi])
        for p, g in zip(params, grads0):
            if g is not None and (p.grad is None or p.grad.data_ptr() != g.data_ptr()):
                p.grad = g
        loss = losses[0] if lanes == 1 else th.stack(losses).mean()
    return loss, flats[0]


def train_lanes(B: int) -> int:
    """Sub-batches the fused training step runs concurrently: 1 unless TQDNE_TRAIN_LANES asks for more (measured at B = 64:
    2 lanes -5 % on the train step = 0.7 % of the bench step, 4 lanes 0; not the default, so that the bench's HIP-event probe
    times B = 64 launches that run alone on their stream)."""
    import os
    env = os.environ.get("TQDNE_TRAIN_LANES")
    if env is not None:
        return max(1, int(env))
    return 1


class _Ctx:
    """stand-in for the autograd context when the loss Function's forward is driven directly"""


def edm_loss(module, sample, eps, unit_noise, cond, cond_sample=None):
    params = [p for p in module.unet.parameters()]
    return _EDMLossFn.apply(module, sample, eps, unit_noise, cond, cond_sample, *params)


class _DenoiseFn(th.autograd.Function):
    """D(x; sigma) = c_skip x + c_out F(c_in x; c_noise) (reference edm.py:105-113) as an ordinary differentiable call: the HIP
    forward keeping what a backward reads (dropout in training mode only), and in backward the hand-written HIP backward seeded with the incoming gradient.
    Differentiable with respect to the UNet parameters (what training code needs) and, when ``sample.requires_grad``, the input
    sample: d D / d x = c_skip + c_out (dF / d x_in) c_in, the middle factor being the stem conv's data gradient."""

    @staticmethod
    def forward(ctx, module, sample, sigma, cond, cond_sample, *params):
        train = module.training
        out = module._denoise_static(sample, sigma, 1, cond, train=train, dropout_seed=rng.next_dropout_seed() if train else 0,
                                     cond_sample=cond_sample)
        ctx.module, ctx.shape, ctx.dev, ctx.lane = module, tuple(sample.shape), sample.device, module._lane
        ctx.concat = cond_sample is not None
        ctx.eng = module.unet._engine(sample.shape[0], sample.shape[2], sample.device, module._lane)
        ctx.fwd_id = ctx.eng._fwd_count
        ctx.want_dx = bool(sample.requires_grad)
        # (c_in / c_skip of THIS call: the module's scalar buffer is overwritten by the next call of the shape)
        ctx.in_skip = module._scalars(sample.shape[0], sample.device)[[0, 2]].clone() if ctx.want_dx else None
        return out.clone()

    @staticmethod
    def backward(ctx, gout):
        module, eng = ctx.module, ctx.eng
        if eng._fwd_count != ctx.fwd_id:
            raise RuntimeError("another forward of the same shape ran between this forward and its backward: the execution plan's "
                               "static buffers no longer hold its activations (call backward before the next forward)")
        B, _, T = ctx.shape
        sc = module._scalars(B, ctx.dev)
        one = th.ones((), device=ctx.dev)
        gout = gout.contiguous().float()
        grads = eng.backward(gout, one, c_out=sc[1], in_scale=None if ctx.concat else sc[0], want_dx=ctx.want_dx)
        dx = None
        if ctx.want_dx:
            c_in, c_skip = ctx.in_skip[0], ctx.in_skip[1]
            dx = eng._bwd.last_dx   # (already times c_in where the stem load applied it; the concatenated input was pre-scaled)
            if ctx.concat:
                dx = dx[:, :ctx.shape[1]] * c_in[:, None, None]
            dx = dx + c_skip[:, None, None] * gout
        return (None, dx, None, None, None) + tuple(grads)


def denoise_with_grad(module, sample, sigma, cond, cond_sample=None):
    """``LightningEDM.forward`` under autograd (train mode, grad enabled): see _DenoiseFn."""
    for name, t in (("sigma", sigma), ("cond", cond), ("cond_sample", cond_sample)):
        if t is not None and t.requires_grad:
            raise NotImplementedError(f"the HIP backward has no gradient with respect to {name}; detach it")
    params = list(module.unet.parameters())
    return _DenoiseFn.apply(module, sample, sigma, cond, cond_sample, *params)


class _UNetFn(th.autograd.Function):
    """``UNetModel.forward`` (reference unet.py:360-398) as an ordinary differentiable call: HIP forward (dropout in training mode
    only) keeping what a backward reads, HIP backward seeded with the incoming gradient.  Gradients: every parameter, and the input
    ``x`` when it requires one (the stem conv's data gradient); ``timesteps`` / ``cond`` have none (no reference caller differentiates
    them) and raise when they ask for one."""

    @staticmethod
    def forward(ctx, model, x, timesteps, cond, *params):
        train = model.training
        eng = model._engine(x.shape[0], x.shape[2], x.device)
        y = eng.forward(x, timesteps, cond, train=train, dropout_seed=rng.next_dropout_seed() if train else 0)
        ctx.eng, ctx.fwd_id, ctx.want_dx = eng, eng._fwd_count, bool(x.requires_grad)
        return y.clone()

    @staticmethod
    def backward(ctx, gy):
        eng = ctx.eng
        if eng._fwd_count != ctx.fwd_id:
            raise RuntimeError("another forward of the same shape ran between this forward and its backward: the execution plan's "
                               "static buffers no longer hold its activations (call backward before the next forward)")
        one = th.ones((), device=gy.device)
        grads = eng.backward(gy.contiguous().float(), one, want_dx=ctx.want_dx)
        return (None, eng._bwd.last_dx if ctx.want_dx else None, None, None) + tuple(grads)


def unet_with_grad(model, x, timesteps, cond):
    """``UNetModel.forward`` with grad enabled: see _UNetFn."""
    for name, t in (("timesteps", timesteps), ("cond", cond)):
        if t is not None and t.requires_grad:
            raise NotImplementedError(f"the HIP backward has no gradient with respect to {name}; detach it")
    return _UNetFn.apply(model, x, timesteps, cond, *model.parameters())
