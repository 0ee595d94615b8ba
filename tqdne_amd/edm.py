"""EDM diffusion wrapper: drop-in for ``tqdne.edm`` (reference tqdne/edm.py:9-251).

``EDM`` carries the constants and scalar maps; ``LightningEDM`` keeps the reference's constructor, attributes and
method names (``forward``, ``step``, ``training_step``, ``validation_step``, ``sample``,
``sample_deterministically``, ``sample_stochastically``, ``evaluate``, ``configure_optimizers``) and runs them on
the HIP kernels:

  forward   edm_scalars -> UNet with c_in folded into the stem load and c_out/c_skip folded into the head epilogue
  step      noise-inject kernel -> forward -> weighted-MSE kernel (loss + dL/dpred) -> hand-written backward
  sample    fp64 state, fp32 network; per step one Euler and one Heun-correction elementwise fp64 kernel;
            sigma stays on the device (no host sync inside the loop)
"""

from __future__ import annotations

import os

import math
from typing import Optional

import torch as th

from . import _lib, engine
from ._cache import scratch_cache
from ._lib import check
from .lightning_compat import LightningModule
from .unet import UNetModel


def _append_dims(x, target_dims):  # reference nn.py:78-83
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


class EDM:
    """Constants and scalar maps of Karras et al. as used by the reference (edm.py:9-52)."""

    sigma_min: float = 0.002
    sigma_max: float = 80.0
    rho: float = 7.0
    sigma_data: float = 0.5
    P_mean: float = -1.2
    P_std: float = 1.2
    S_churn: float = 40
    S_min: float = 0.05
    S_max: float = 50
    S_noise: float = 1.003

    def sigma(self, eps):
        return (eps * self.P_std + self.P_mean).exp()

    def loss_weight(self, sigma):
        return (sigma**2 + self.sigma_data**2) / (sigma * self.sigma_data) ** 2

    def skip_scaling(self, sigma):
        return self.sigma_data**2 / (sigma**2 + self.sigma_data**2)

    def out_scaling(self, sigma):
        return sigma * self.sigma_data / (sigma**2 + self.sigma_data**2) ** 0.5

    def in_scaling(self, sigma):
        return 1 / (sigma**2 + self.sigma_data**2) ** 0.5

    def noise_conditioning(self, sigma):
        return 0.25 * sigma.log()

    def sampling_sigmas(self, num_steps, device=None):
        inv = 1 / self.rho
        idx = th.arange(num_steps, dtype=th.float32, device=device)
        s = (self.sigma_max**inv + idx / (num_steps - 1) * (self.sigma_min**inv - self.sigma_max**inv)) ** self.rho
        return th.cat([s, th.zeros_like(s[:1])])

    def sigma_hat(self, sigma, num_steps):
        gamma = min(self.S_churn / num_steps, 2**0.5 - 1) if self.S_min <= sigma <= self.S_max else 0
        return sigma + gamma * sigma


def sampler_lanes(B: int) -> int:
    """Sub-batches (HIP streams) the deterministic sampler integrates concurrently: up to 4, at least 16 samples each.
    TQDNE_SAMPLER_LANES overrides (1 = one stream)."""
    env = os.environ.get("TQDNE_SAMPLER_LANES")
    if env is not None:
        return max(1, int(env))
    for n in (4, 2):
        if B % n == 0 and B // n >= 16:
            return n
    return 1


def _p(t):
    return None if t is None else t.data_ptr()


class LightningEDM(LightningModule):
    def __init__(
        self,
        unet_config: dict,
        optimizer_params: dict,
        num_sampling_steps: int = 25,
        deterministic_sampling: bool = True,
        edm: EDM = EDM(),
        autoencoder=None,
    ):
        super().__init__()
        self.unet = UNetModel(**unet_config)
        self.optimizer_params = optimizer_params
        self.num_sampling_steps = num_sampling_steps
        self.deterministic_sampling = deterministic_sampling
        self.edm = edm
        self.autoencoder = autoencoder.eval() if autoencoder else None
        self.config = unet_config
        if self.autoencoder:
            for param in self.autoencoder.parameters():
                param.requires_grad = False
        self.save_hyperparameters(ignore=("autoencoder"))
        self._scal = scratch_cache()
        self._lane = 0  # which set of static buffers / execution plan the calls below use (two-lane sampling)

    # ------------------------------------------------------------------ preconditioned network
    def _scalars(self, B, device):
        key = (B, str(device), self._lane)
        s = self._scal.get(key)
        if s is None:
            s = th.empty(5, B, dtype=th.float32, device=device)  # c_in, c_out, c_skip, c_noise, loss weight
            self._scal[key] = s
        return s

    def _denoise_static(self, sample, sigma, sigma_stride, cond, train=False, dropout_seed=0, cond_sample=None, infer=False):
        """Fused preconditioned forward (edm.py:105-113); returns the engine's static output buffer.
        ``sigma``: device tensor; ``sigma_stride`` 1 (per-sample) or 0 (one value shared by the batch).
        ``cond_sample``: conditioning signal concatenated on the channel axis behind the scaled sample (edm.py:108-109); the
        pre-scale then cannot ride in the stem load (only the first channels are scaled): one fused concat + scale launch."""
        lib = _lib.load()
        B, _, T = sample.shape
        dev = sample.device
        sc = self._scalars(B, dev)
        stream = th.cuda.current_stream(dev).cuda_stream
        check(lib.tq_edm_scalars(_p(sigma), sigma_stride, float(self.edm.sigma_data), _p(sc[0]), _p(sc[1]), _p(sc[2]),
                                 _p(sc[3]), _p(sc[4]), B, stream), "edm scalars")
        if cond_sample is not None:
            C1 = cond_sample.shape[1]
            key = ("cat", B, sample.shape[1], C1, T, str(dev), self._lane)
            x_in = self._scal.get(key)
            if x_in is None:
                x_in = th.empty(B, sample.shape[1] + C1, T, dtype=th.float32, device=dev)
                self._scal[key] = x_in
            check(lib.tq_concat_scale(_p(sample), _p(sc[0]), _p(cond_sample.contiguous().float()), _p(x_in), B, sample.shape[1], C1, T,
                                      stream), "concat + scale")
            eng = self.unet._engine(B, T, dev, self._lane)
            return eng.forward(x_in, sc[3], cond, in_scale=None, c_out=sc[1], c_skip=sc[2], skip_src=sample, train=train,
                               dropout_seed=dropout_seed, infer=infer)
        eng = self.unet._engine(B, T, dev, self._lane)
        return eng.forward(sample, sc[3], cond, in_scale=sc[0], c_out=sc[1], c_skip=sc[2], skip_src=sample, train=train,
                           dropout_seed=dropout_seed, infer=infer)

    def forward(self, sample, sigma, cond_sample=None, cond=None):
        """Make a forward pass through the network with skip connection (edm.py:105-113)."""
        if self.unet.dims == 2:   # generate_waveforms.py family: stock PyTorch operators, differentiable by torch.autograd
            from . import family2d
            return family2d.denoise(self, sample, sigma, cond_sample, cond)
        engine.require_device(sample)
        sample = sample.contiguous()
        sigma = sigma.contiguous().float()
        assert (cond is not None) == (self.unet.cond_features is not None), (
            "must specify cond if and only if the model is conditioned"
        )
        if cond_sample is not None:
            cond_sample = cond_sample.contiguous()
        if th.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in self.unet.parameters())):
            # an ordinary differentiable call, as in the reference: parameter gradients, and d / d sample when the sample asks for it
            # (dropout only in training mode)
            from .autograd import denoise_with_grad
            return denoise_with_grad(self, sample, sigma, cond, cond_sample)
        y = self._denoise_static(sample, sigma, 1, cond, cond_sample=cond_sample, infer=True).clone()
        if self.unet._engine(sample.shape[0], sample.shape[2], sample.device, self._lane).check_range():
            y = self._denoise_static(sample, sigma, 1, cond, cond_sample=cond_sample, infer=True).clone()  # (now on bf16x3)
        return y

    # ------------------------------------------------------------------ training
    def step(self, batch, batch_idx):
        """A single step in the training loop (edm.py:115-134)."""
        sample = batch["signal"]
        cond_sample = batch["cond_signal"] if "cond_signal" in batch else None
        cond = batch["cond"] if "cond" in batch else None
        if self.autoencoder:
            sample = self.autoencoder.encode(sample)
            if cond_sample is not None:
                cond_sample = self.autoencoder.encode(cond_sample)
        eps = th.randn(sample.shape[0], device=self.device)
        unit_noise = th.randn_like(sample)
        return self.step_with_noise(sample, eps, unit_noise, cond=cond, cond_sample=cond_sample)

    def step_with_noise(self, sample, eps, unit_noise, cond=None, cond_sample=None):
        """``step`` with the two random draws of edm.py:126,128 supplied by the caller (tests inject CPU draws)."""
        if self.unet.dims == 2:
            from . import family2d
            return family2d.edm_loss(self, sample, eps, unit_noise, cond, cond_sample)
        from .autograd import edm_loss
        return edm_loss(self, sample.contiguous(), eps.contiguous().float(), unit_noise.contiguous(), cond,
                        None if cond_sample is None else cond_sample.contiguous())

    def step_and_backward(self, batch, on_bucket=None, bucket_elems: int = 4 << 20, tail_fill=None):
        """``step`` + backward in one call, gradients left in ``p.grad`` (views of one flat buffer, returned as well).
        ``on_bucket``: gradient-exchange hook, called as buckets of the flat buffer become final (BackwardPlan.run);
        ``tail_fill``: extra words of the caller that ride at the end of the last bucket (BackwardPlan.run)."""
        from .autograd import edm_loss_and_grads
        if self.unet.dims == 2:
            raise NotImplementedError("DataParallelTrainer drives the 1-D HIP path; train dims=2 models with step() + torch.autograd")
        sample = batch["signal"]
        cond = batch["cond"] if "cond" in batch else None
        cond_sample = batch["cond_signal"] if "cond_signal" in batch else None
        if self.autoencoder:
            with th.no_grad():
                sample = self.autoencoder.encode(sample)
                if cond_sample is not None:
                    cond_sample = self.autoencoder.encode(cond_sample)
        eps = th.randn(sample.shape[0], device=sample.device)
        unit_noise = th.randn_like(sample)
        return edm_loss_and_grads(self, sample.contiguous(), eps, unit_noise, cond,
                                  None if cond_sample is None else cond_sample.contiguous(), on_bucket=on_bucket,
                                  bucket_elems=bucket_elems, tail_fill=tail_fill)

    def training_step(self, batch, batch_idx):
        loss = self.step(batch, batch_idx)
        self.log("training/loss", loss.item(), sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self.step(batch, batch_idx)
        self.log("validation/loss", loss.item(), sync_dist=True)
        return loss

    # ------------------------------------------------------------------ sampling
    @th.no_grad()
    def sample(self, shape, cond_sample=None, cond=None):
        """Sample using Heun's second order method (edm.py:146-169)."""
        dtype = th.float64
        if self.autoencoder:
            if cond_sample is not None:
                cond_sample = self.autoencoder.encode(cond_sample)
            # the reference encodes a zeros tensor just to learn the latent shape (edm.py:154-157); the shape is known
            # in closed form, so the wasted encoder pass is skipped
            enc = self.autoencoder.encoder
            shape = (shape[0], enc.out_channels // 2) + tuple(n // enc.time_scale for n in shape[2:])
        # schedule built on the host in fp32 exactly like the reference's CPU path, then moved (pow differs by ulps on device)
        sigmas = self.edm.sampling_sigmas(self.num_sampling_steps).to(self.device)
        eps = th.randn(shape, device=self.device, dtype=dtype) * sigmas[0]
        if self.unet.dims == 2:
            from . import family2d
            churn = None if self.deterministic_sampling else th.randn_like
            sample = family2d.heun_sample(self, eps, sigmas, cond_sample, cond, churn).to(th.float32)
            return self.autoencoder.decode(sample) if self.autoencoder else sample
        if self.deterministic_sampling:
            sample = self.sample_deterministically(eps, sigmas, cond_sample, cond)
        else:
            sample = self.sample_stochastically(eps, sigmas, cond_sample, cond)
        sample = sample.to(th.float32)
        if self.autoencoder:
            return self.autoencoder.decode(sample)
        return sample

    def _sampler_buffers(self, eps):
        key = ("smp", tuple(eps.shape), str(eps.device), self._lane)
        bufs = self._scal.get(key)
        if bufs is None:
            f64 = lambda: th.empty(eps.shape, dtype=th.float64, device=eps.device)
            bufs = dict(x=f64(), xn=f64(), d=f64(), x32=th.empty(eps.shape, dtype=th.float32, device=eps.device))
            self._scal[key] = bufs
        return bufs

    @th.no_grad()
    def sample_deterministically(self, eps, sigmas, cond_sample=None, cond=None, use_graph=None, lanes=None):
        """Deterministic Heun sampler (edm.py:171-196): ``eps`` is the fp64 start state (already scaled by sigmas[0]),
        ``sigmas`` the fp32 schedule ending in 0.  NFE = 2*len(sigmas) - 3 when the last sigma is the appended 0.

        ``lanes``: samples are independent, so the batch can be integrated as ``lanes`` sub-batches on as many HIP streams, each
        with its own execution plan.  Workgroups of one launch run in lockstep (all in their load prologue, then all in their MFMA loop,
        then all in their store epilogue); streams drift out of phase, so one lane's prologue / epilogue bursts overlap another
        lane's matrix work (measured on single layers: 1.0-1.17x, tools/desync_bench.py; 18-step sample at B = 64: 2 lanes
        -5.5 %, 4 lanes -9 %, 8 lanes +16 %: launches too small and too many).  Results are bit-identical to one lane.
        Default: ``sampler_lanes(B)``; 1 under graph replay.

        ``use_graph``: True replays the WHOLE integration (every network evaluation and every fp64 update of all steps) from one
        HIP graph captured on first use -- one host call per sample instead of ~100 launches per network evaluation; "denoiser" is
        round 1's form (one captured network evaluation, replayed per NFE); False launches eagerly.  Default (None): eager, unless
        TQDNE_SAMPLER_GRAPH=1.  Opt-in since round 4: measured neutral even for launch-bound batches (tiny UNet, B = 4: the GPU-side
        kernel boundaries bound the loop, not the host), and a capture on first use inside a validation loop is exposed to whatever
        other threads do with the device meanwhile (the capture is thread-local for that reason)."""
        if not eps.is_cuda:
            raise RuntimeError("tqdne_amd samples on MI355X HIP kernels only; got a CPU start state")
        if use_graph is None:
            use_graph = os.environ.get("TQDNE_SAMPLER_GRAPH") == "1" and lanes is None and not th.cuda.is_current_stream_capturing()
        out = self._sample_det(eps, sigmas, cond_sample, cond, use_graph, lanes)
        # range guard of the fp16-range conv scheme: one flag read per sample call; if a tensor came near the fp16 range the plans
        # have been moved to bf16x3 and the integration is repeated
        # (the flag is one per model and device, shared by every lane's plan: one read)
        engs = [e for e in self.unet._engine_cache.values() if e.dev == eps.device]
        if engs and engs[0].check_range():
            out = self._sample_det(eps, sigmas, cond_sample, cond, use_graph, lanes)
        return out

    def _sample_det(self, eps, sigmas, cond_sample, cond, use_graph, lanes):
        B = eps.shape[0]
        if lanes is None:
            lanes = sampler_lanes(B)
        if use_graph:
            lanes = 1  # one captured denoiser per buffer set; replay is for launch-bound (small) batches, where lanes do not pay
        if use_graph is True:
            return self._graph_sample(eps, sigmas, cond_sample, cond)
        use_graph = bool(use_graph)   # ("denoiser": the per-evaluation graph below)
        if lanes < 2 or B % lanes or B // lanes < 8:
            run = self._heun_lane(eps, sigmas, cond_sample, cond, use_graph)
            for _ in run:
                pass
            res = run.result.clone()
            run.release()
            return res
        return self._run_lanes(eps, sigmas, cond_sample, cond, lanes, use_graph)

    def _run_lanes(self, eps, sigmas, cond_sample, cond, lanes, use_graph=False, churn=None):
        """``lanes`` sub-batches integrated concurrently on as many HIP streams (see sample_deterministically); ``churn``: the
        stochastic sampler's (sigma_hat, coefficients, unit noises per step for the WHOLE batch) -- every lane takes its slice."""
        B = eps.shape[0]
        dev = eps.device
        h = B // lanes
        cut = lambda t, i: None if t is None else t[i * h:(i + 1) * h].contiguous()
        main = th.cuda.current_stream(dev)
        streams = [main] + [self._side_stream(dev, i) for i in range(1, lanes)]
        for st in streams[1:]:
            st.wait_stream(main)
        runs = []
        try:
            for i, st in enumerate(streams):
                self._lane = engine.CONCURRENT_LANE0 + i
                with th.cuda.stream(st):
                    lane_churn = None
                    if churn is not None:
                        shat, coef, noises = churn
                        lane_churn = (shat, coef, lambda k, _i=i, _st=st: noises.take(k, _i, h, _st))
                    runs.append(self._heun_lane(cut(eps, i), sigmas, cut(cond_sample, i), cut(cond, i), use_graph, churn=lane_churn))
            # one sampler step of lane 0, then of lane 1, ...: all queues stay fed well ahead of the GPU
            while not all(r.done for r in runs):
                for i, (r, st) in enumerate(zip(runs, streams)):
                    if not r.done:
                        self._lane = engine.CONCURRENT_LANE0 + i
                        with th.cuda.stream(st):
                            r.advance()
        finally:
            self._lane = 0
        out = th.empty_like(eps)
        for i, st in enumerate(streams[1:], 1):
            with th.cuda.stream(st):
                out[i * h:(i + 1) * h].copy_(runs[i].result)
            main.wait_stream(st)
        out[:h].copy_(runs[0].result)
        for r in runs:
            r.release()
        return out

    class _StepNoises:
        """The churned sampler's unit noises, one (B, C, T) fp64 draw per step for the WHOLE batch, handed out per lane.  Step k is drawn
        when the FIRST lane asks for it -- the lanes advance step-major (``_run_lanes``), so this is lane 0 on the caller's stream and
        the draws come in the one-lane loop's order (edm.py:207's ``randn_like`` per step) -- and let go once the last lane has taken
        its slice: at most a step or two are alive instead of all ``num_sampling_steps`` (0.45 GB at B = 256; round-5 advisor).
        Stream safety: the draw is recorded as an event that every other lane's stream waits for, and each slice is registered with
        the lane's stream (``record_stream``), so the allocator does not hand the block out again before the lane's launches ran."""

        def __init__(self, shape, dev, lanes, given=None):
            self.shape, self.dev, self.lanes, self.given = tuple(shape), dev, lanes, given
            self.live = {}   # step -> [tensor, event, slices still to hand out]
            self.drawn = 0

        def take(self, k, lane, h, stream):
            ent = self.live.get(k)
            if ent is None:
                if self.given is not None:
                    t = self.given[k].to(device=self.dev, dtype=th.float64).contiguous()
                else:
                    assert k == self.drawn, "noises are drawn in step order"
                    t = th.randn(self.shape, dtype=th.float64, device=self.dev)
                self.drawn = max(self.drawn, k + 1)
                ev = th.cuda.Event()
                ev.record(th.cuda.current_stream(self.dev))
                ent = self.live[k] = [t, ev, self.lanes, th.cuda.current_stream(self.dev)]
            t, ev, left, src = ent
            if stream != src:
                stream.wait_event(ev)
                t.record_stream(stream)
            ent[2] = left - 1
            if ent[2] == 0:
                del self.live[k]
            return t[lane * h:(lane + 1) * h]

    def _side_stream(self, dev, i=1):
        return engine.side_stream(dev, i)   # (one pool per process: the number of live streams matters, see engine.side_stream)

    def _heun_lane(self, eps, sigmas, cond_sample, cond, use_graph, churn=None):
        """The Heun integration of one (half) batch as a resumable object: ``advance()`` enqueues one sampler step on the current
        stream with the current lane's plan and buffers; ``result`` is the fp64 state buffer once ``done``.
        ``churn`` = (sigma_hat (steps,), coefficients (steps,), noise_of_step): the stochastic sampler's steps (edm.py:198-230) -- the
        noise increase ``tq_heun_churn`` with this lane's slice ``noise_of_step(i)`` of the step's unit noise, then the Heun step from
        sigma_hat instead of sigma."""
        lib = _lib.load()
        dev = eps.device
        sigmas = sigmas.to(device=dev, dtype=th.float32).contiguous()
        if cond is not None:
            cond = cond.contiguous().float()
        if cond_sample is not None:
            cond_sample = cond_sample.contiguous().float()
        bufs = self._sampler_buffers(eps)
        edm = self

        class _Run:
            def __init__(r, start):
                # (``start`` is an argument, not a closure variable: a class object sits in reference cycles of its own, and a
                # cell holding a view of the caller's start state would keep that whole allocation alive until a cyclic GC pass)
                r.x, r.xn, r.d, r.x32 = bufs["x"], bufs["xn"], bufs["d"], bufs["x32"]
                if churn is not None:
                    if "xh" not in bufs:
                        bufs["xh"] = th.empty_like(bufs["x"])
                    r.xh = bufs["xh"]
                r.x.copy_(start)
                r.x32.copy_(start)  # fp64 -> fp32 rounding, as sample_curr.to(self.dtype)
                r.i, r.nsteps = 0, sigmas.numel() - 1
                r.keep = (sigmas, cond, cond_sample)
                if use_graph:
                    r.denoise = edm._graph_denoiser(bufs, r.x32, cond, cond_sample)
                else:
                    r.denoise = lambda sig_ptr: edm._denoise_static(r.x32, _RawPtr(sig_ptr), 0, cond, cond_sample=cond_sample, infer=True)

            @property
            def done(r):
                return r.i >= r.nsteps

            @property
            def result(r):
                return r.x

            def advance(r):
                i, n, sp = r.i, r.x.numel(), sigmas.data_ptr()
                stream = th.cuda.current_stream(dev).cuda_stream
                s_i, s_n = sp + 4 * i, sp + 4 * (i + 1)
                if churn is not None:
                    shat, coef, noise_of_step = churn
                    unit = noise_of_step(i)   # (a slice registered with this lane's stream: safe to let go once the launch is enqueued)
                    s_hat = shat.data_ptr() + 4 * i
                    check(lib.tq_heun_churn(_p(r.x), _p(unit), coef.data_ptr() + 4 * i, float(edm.edm.S_noise), _p(r.xh), _p(r.x32), n,
                                            stream), "heun churn")
                    den = r.denoise(s_hat)
                    check(lib.tq_heun_euler(_p(r.xh), _p(den), s_hat, s_n, _p(r.d), _p(r.xn), _p(r.x32), n, stream), "heun euler")
                    if i < edm.num_sampling_steps - 1:
                        den = r.denoise(s_n)
                        check(lib.tq_heun_correct(_p(r.xh), _p(r.xn), _p(den), _p(r.d), s_hat, s_n, _p(r.x), _p(r.x32), n, stream),
                              "heun correct")
                    else:
                        r.x, r.xn = r.xn, r.x
                    r.i += 1
                    return
                den = r.denoise(s_i)
                check(lib.tq_heun_euler(_p(r.x), _p(den), s_i, s_n, _p(r.d), _p(r.xn), _p(r.x32), n, stream), "heun euler")
                if i < edm.num_sampling_steps - 1:
                    den = r.denoise(s_n)
                    check(lib.tq_heun_correct(_p(r.x), _p(r.xn), _p(den), _p(r.d), s_i, s_n, _p(r.x), _p(r.x32), n, stream),
                          "heun correct")
                else:
                    r.x, r.xn = r.xn, r.x
                r.i += 1

            def __iter__(r):
                while not r.done:
                    r.advance()
                    yield r.i

            def release(r):
                """drop the references that tie this object into a cycle (closure <-> instance): the start state and the
                conditioning tensors would otherwise stay allocated until Python's cyclic collector gets round to it"""
                r.keep = r.denoise = None

        run = _Run(eps)
        del eps
        return run

    def _graph_sample(self, eps, sigmas, cond_sample, cond):
        """The whole Heun integration as ONE HIP graph.  Everything the captured launches read lives in static buffers of the sampler
        (start state, sigma schedule, conditioning), refreshed by device-to-device copies before each replay; the graph is re-captured
        when the shapes, the number of steps or the plan (weights format, see the range guard) change."""
        dev = eps.device
        B = eps.shape[0]
        bufs = self._sampler_buffers(eps)
        eng = self.unet._engine(B, eps.shape[2], dev, self._lane)
        nsig = int(sigmas.numel())
        key = (nsig, None if cond is None else tuple(cond.shape), None if cond_sample is None else tuple(cond_sample.shape),
               eng.uid, eng.plan_epoch, self.num_sampling_steps)
        cache = bufs.setdefault("loop_graphs", {})   # one captured loop per key (a sweep over step counts re-uses them)
        live = {e.uid for e in self.unet._engine_cache.values()}
        for k_ in [k_ for k_ in cache if k_[3] not in live]:   # graphs of evicted plans: their launches point into freed buffers
            del cache[k_]
        g = cache.get(key)
        if g is None:
            st = dict(key=key, sig=th.empty(nsig, dtype=th.float32, device=dev), start=th.empty_like(bufs["x"]),
                      cond=None if cond is None else th.empty(cond.shape, dtype=th.float32, device=dev),
                      cs=None if cond_sample is None else th.empty(cond_sample.shape, dtype=th.float32, device=dev))
            st["sig"].copy_(sigmas)
            st["start"].copy_(eps)
            if cond is not None:
                st["cond"].copy_(cond)
            if cond_sample is not None:
                st["cs"].copy_(cond_sample)
            # warm-up outside capture: plan build, weight packing, allocator state
            run = self._heun_lane(st["start"], st["sig"], st["cs"], st["cond"], False)
            run.advance()
            run.release()
            th.cuda.synchronize(dev)
            graph = th.cuda.CUDAGraph()
            # thread-local capture: other threads of the process (a DataLoader's pin-memory thread, RCCL's watchdog) may allocate
            # or record events meanwhile without invalidating it
            with th.cuda.graph(graph, capture_error_mode="thread_local"):
                run = self._heun_lane(st["start"], st["sig"], st["cs"], st["cond"], False)
                for _ in run:
                    pass
                st["out"] = run.result
                run.release()
            st["graph"] = graph
            while len(cache) >= 4:   # (each graph owns a private pool: keep a handful, drop the oldest)
                cache.pop(next(iter(cache)))
            cache[key] = g = st
        # the captured launches read the model's packed weight fragments at fixed addresses: bring them up to date with the parameters
        # (an optimizer step since the last call) BEFORE the replay -- inside a forward this is the first thing eng.forward does
        stream = th.cuda.current_stream(dev).cuda_stream
        eng.repack(stream)
        g["sig"].copy_(sigmas)
        g["start"].copy_(eps)
        if cond is not None:
            g["cond"].copy_(cond)
        if cond_sample is not None:
            g["cs"].copy_(cond_sample)
        g["graph"].replay()
        eng._mark_use(stream)   # (a repack issued on another stream must wait for this replay's reads of the fragments)
        return g["out"].clone()

    def _graph_denoiser(self, bufs, x32, cond, cond_sample=None):
        """One preconditioned UNet evaluation (~160 launches) captured once in a HIP graph and replayed per NFE; sigma is fed
        through a static device slot.  Pays off when the forward is launch-bound (small batches); at B = 64 the host already
        runs ahead of the GPU."""
        g = bufs.get("graph")
        eng = self.unet._engine(x32.shape[0], x32.shape[2], x32.device, self._lane)
        key = (None if cond is None else cond.data_ptr(), None if cond_sample is None else cond_sample.data_ptr(), eng.uid, eng.plan_epoch)
        if g is None or bufs.get("graph_cond") != key:
            slot = th.ones(1, device=x32.device)  # (a valid sigma for the warm-up: sigma = 0 gives c_noise = -inf, NaN activations)
            self._denoise_static(x32, slot, 0, cond, cond_sample=cond_sample, infer=True)  # warm-up outside capture (plan build, packing)
            th.cuda.synchronize(x32.device)
            graph = th.cuda.CUDAGraph()
            with th.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self._denoise_static(x32, slot, 0, cond, cond_sample=cond_sample, infer=True)
            g = (graph, slot, out)
            bufs["graph"], bufs["graph_cond"] = g, key
        graph, slot, out = g
        elem = slot.element_size()

        def run(sig_ptr):
            # (packed weights first: see _graph_sample) device-to-device copy of the 4-byte sigma into the captured slot, then replay
            stream = th.cuda.current_stream(slot.device).cuda_stream
            eng.repack(stream)
            _lib_memcpy_d2d(slot.data_ptr(), sig_ptr, elem, stream)
            graph.replay()
            eng._mark_use(stream)
            return out

        return run

    @th.no_grad()
    def sample_stochastically(self, eps, sigmas, cond_sample=None, cond=None, churn_noises=None, lanes=None):
        """Stochastic (churned) sampler (edm.py:198-230) on the HIP kernels: per step the noise increase ``tq_heun_churn`` (lines
        205-208), the fused denoiser at sigma_hat, ``tq_heun_euler`` and -- except on the last step -- the denoiser at sigma_next
        and ``tq_heun_correct`` (lines 210-228), fp64 state, fp32 network.  sigma_hat and sqrt(sigma_hat^2 - sigma^2) are formed once
        on the host from the reference's own fp32 0-dim-tensor expressions (edm.py:48-52, 208) and live on the device; the only torch
        op in the loop is the ``randn_like`` draw of line 207 (``churn_noises[i]``, fp64 unit draws, replace it in tests)."""
        if not eps.is_cuda:
            raise RuntimeError("tqdne_amd samples on MI355X HIP kernels only; got a CPU start state")
        lib = _lib.load()
        dev = eps.device
        N = self.num_sampling_steps
        sig_cpu = sigmas.detach().to("cpu", th.float32)
        shat_cpu = th.stack([self.edm.sigma_hat(sg, N) for sg in sig_cpu[:-1]]).to(th.float32)
        coef_cpu = th.stack([(sh ** 2 - sg ** 2) ** 0.5 for sh, sg in zip(shat_cpu, sig_cpu[:-1])]).to(th.float32)
        sig = sig_cpu.to(dev).contiguous()
        shat, coef = shat_cpu.to(dev).contiguous(), coef_cpu.to(dev).contiguous()
        if cond is not None:
            cond = cond.contiguous().float()
        if cond_sample is not None:
            cond_sample = cond_sample.contiguous().float()
        B = eps.shape[0]
        if lanes is None:
            lanes = sampler_lanes(B)
        if lanes >= 2 and B % lanes == 0 and B // lanes >= 8:
            # round 5: the stochastic sampler on the lanes of the deterministic one.  A step's unit noise is drawn for the whole batch,
            # in the order the one-lane loop draws them (line 207's randn_like per step), when the first lane reaches the step, and cut
            # per lane (_StepNoises): the result does not depend on the number of lanes (>= 2), and equals the one-lane integration of
            # the same draws bit for bit where the one-lane plan uses the same tiles (it does at B = 64; a small solo batch takes the
            # small position tile on some levels, engine.SMALL_TILE_WGS, which associates the GroupNorm sums differently: ~1e-6).
            rng_state = th.cuda.get_rng_state(dev) if churn_noises is None else None
            noises = self._StepNoises(eps.shape, dev, lanes, churn_noises)
            out = self._run_lanes(eps, sig, cond_sample, cond, lanes, churn=(shat, coef, noises))
            engs = [e for e in self.unet._engine_cache.values() if e.dev == dev]
            if engs and engs[0].check_range():   # (the plans are on bf16x3 now: the same draws again)
                if rng_state is not None:
                    th.cuda.set_rng_state(rng_state, dev)
                return self.sample_stochastically(eps, sigmas, cond_sample, cond, churn_noises, lanes)
            return out
        bufs = self._sampler_buffers(eps)
        if "xh" not in bufs:
            bufs["xh"] = th.empty_like(bufs["x"])
        x, xh, xn, d, x32 = bufs["x"], bufs["xh"], bufs["xn"], bufs["d"], bufs["x32"]
        x.copy_(eps)
        n = x.numel()
        stream = th.cuda.current_stream(dev).cuda_stream
        nsteps = sig.numel() - 1
        for i in range(nsteps):
            unit = th.randn_like(x) if churn_noises is None else churn_noises[i].to(device=dev, dtype=th.float64).contiguous()
            s_hat, s_next = shat.data_ptr() + 4 * i, sig.data_ptr() + 4 * (i + 1)
            check(lib.tq_heun_churn(_p(x), _p(unit), coef.data_ptr() + 4 * i, float(self.edm.S_noise), _p(xh), _p(x32), n, stream),
                  "heun churn")
            den = self._denoise_static(x32, _RawPtr(s_hat), 0, cond, cond_sample=cond_sample, infer=True)
            check(lib.tq_heun_euler(_p(xh), _p(den), s_hat, s_next, _p(d), _p(xn), _p(x32), n, stream), "heun euler")
            if i < N - 1:
                den = self._denoise_static(x32, _RawPtr(s_next), 0, cond, cond_sample=cond_sample, infer=True)
                check(lib.tq_heun_correct(_p(xh), _p(xn), _p(den), _p(d), s_hat, s_next, _p(x), _p(x32), n, stream), "heun correct")
            else:
                x, xn = xn, x
        out = x.clone()
        if self.unet._engine(eps.shape[0], eps.shape[2], dev, self._lane).check_range():
            return self.sample_stochastically(eps, sigmas, cond_sample, cond, churn_noises)  # (the plan is on bf16x3 now)
        return out

    @th.no_grad()
    def evaluate(self, batch):
        """Evaluate the model on a batch of data (edm.py:232-238)."""
        sample = batch["signal"]
        cond_sample = batch["cond_signal"] if "cond_signal" in batch else None
        cond = batch["cond"] if "cond" in batch else None
        return self.sample(sample.shape, cond_sample, cond)

    def configure_optimizers(self):
        optimizer = th.optim.Adam(self.parameters(), lr=self.optimizer_params["learning_rate"])
        lr_scheduler = th.optim.lr_scheduler.CosineAnnealingLR(
            optimizer, T_max=self.optimizer_params["max_steps"], eta_min=self.optimizer_params["eta_min"]
        )
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": lr_scheduler, "interval": "step"}}


def _lib_memcpy_d2d(dst, src, nbytes, stream):
    """async device-to-device copy on ``stream`` (hipMemcpyAsync through the runtime torch already loaded)"""
    import ctypes
    rt = _hip_runtime()
    rc = rt.hipMemcpyAsync(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(nbytes), 3, ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"hipMemcpyAsync failed: {rc}")


_HIP_RT = []


def _hip_runtime():
    if not _HIP_RT:
        import ctypes
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _HIP_RT.append(ctypes.CDLL(name))
                break
            except OSError:
                continue
        if not _HIP_RT:
            raise RuntimeError("libamdhip64.so not found")
    return _HIP_RT[0]


class _RawPtr:
    """A bare device address standing in for a tensor where only ``data_ptr()`` is needed."""

    __slots__ = ("ptr",)

    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr
