"""Consistency model: drop-in for ``tqdne.consistency_model`` (reference tqdne/consistency_model.py:63-190): forward / sample
(63-106) and the iCT training step (115-176: teacher at sigma_t without gradient, student at sigma_{t+1}, weighted pseudo-Huber
distance).  The network is a ``tqdne_amd.UNetModel``; the consistency skip/out scalings are folded into the head-conv epilogue,
and the raw sigma is the network's timestep (line 77).  Both UNet passes of the training step and its backward are HIP."""

from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib, engine, rng
from ._lib import check
from ._cache import scratch_cache
from .edm import sampler_lanes
from .lightning_compat import LightningModule


def _p(t):
    return None if t is None else t.data_ptr()




class _ICTLossFn(torch.autograd.Function):
    """loss = mean(w_t * (sqrt((f(x + s_{t+1} eps, s_{t+1}) - sg[f(x + s_t eps, s_t)])^2 + c^2) - c)),  c = 0.00054 sqrt(dim)."""

    @staticmethod
    def forward(ctx, module, sample, sigmas, timesteps, epsilon, cond, *params):
        engine.require_device(sample)
        B, nd = sample.shape[0], sample.dim()
        seed = rng.next_dropout_seed()  # one seed: teacher = student masks
        train = module.training
        lib = _lib.load()
        dev = sample.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        t_sig, s_sig = sigmas[timesteps].float().contiguous(), sigmas[timesteps + 1].float().contiguous()
        per = sample[0].numel()
        key = ("ict", tuple(sample.shape), str(dev))
        bufs = module._scal.get(key)
        if bufs is None:
            bufs = dict(xt=torch.empty_like(sample), xs=torch.empty_like(sample), target=torch.empty_like(sample),
                        dpred=torch.empty_like(sample), loss=torch.empty(1, device=dev))
            module._scal[key] = bufs
        epsilon = epsilon.contiguous()
        # the two noised copies (consistency_model.py:150-160), teacher first (no gradient, same dropout masks as the student)
        check(lib.tq_axpy_sigma(_p(sample), _p(epsilon), _p(t_sig), _p(bufs["xt"]), B, per, stream), "noise (teacher)")
        check(lib.tq_axpy_sigma(_p(sample), _p(epsilon), _p(s_sig), _p(bufs["xs"]), B, per, stream), "noise (student)")
        bufs["target"].copy_(module._forward_static(bufs["xt"], t_sig, cond, train=train, dropout_seed=seed))
        pred = module._forward_static(bufs["xs"], s_sig, cond, train=train, dropout_seed=seed)
        c = 0.00054 * float(np.sqrt(np.prod(sample.shape[2:])))
        w = (1 / (sigmas[1:] - sigmas[:-1]))[timesteps].float().contiguous()   # (B,) weights: indexing glue, as the schedule
        check(lib.tq_pseudo_huber_loss(_p(pred), _p(bufs["target"]), _p(w), c, _p(bufs["loss"]), _p(bufs["dpred"]), B, per, stream),
              "pseudo-Huber loss")
        ctx.module, ctx.shape = module, tuple(sample.shape)
        ctx.dpred = bufs["dpred"]
        # (the plan stays with the graph: a later look-up could find a NEW plan if the bounded cache evicted this one in between)
        ctx.eng = module.net._engine(B, sample.shape[2], dev)
        ctx.fwd_id = ctx.eng._fwd_count
        return bufs["loss"][0].clone()

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.eng
        if eng._fwd_count != ctx.fwd_id:
            raise RuntimeError("another forward of the same shape ran between this loss and its backward: the execution plan's static "
                               "buffers no longer hold its activations")
        grads = eng.backward(ctx.dpred, gloss)
        return (None,) * 6 + tuple(grads)


class LithningConsistencyModel(LightningModule):  # (sic) the reference's class name
    def __init__(self, net, sigma_min=0.002, sigma_max=80.0, rho=7.0, sigma_data=0.5, initial_timesteps=10,
                 final_timesteps=1280, lognormal_mean=-1.1, lognormal_std=2.0, lr=1e-4):
        super().__init__()
        if getattr(net, "dims", 1) != 1:
            raise NotImplementedError("the consistency model runs on the 1-D HIP path (every reference config that uses it is 1-D)")
        self.net = net
        self.sigma_min, self.sigma_max, self.rho, self.sigma_data = sigma_min, sigma_max, rho, sigma_data
        self.initial_timesteps, self.final_timesteps = initial_timesteps, final_timesteps
        self.lognormal_mean, self.lognormal_std, self.lr = lognormal_mean, lognormal_std, lr
        self._scal = scratch_cache()

    def _forward_static(self, sample, sigma, cond, lane=0, train=False, dropout_seed=0, infer=False):
        lib = _lib.load()
        B, _, T = sample.shape
        dev = sample.device
        key = (B, str(dev), lane)
        sc = self._scal.get(key)
        if sc is None:
            sc = torch.empty(2, B, device=dev)
            self._scal[key] = sc
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib.tq_cm_scalars(_p(sigma), 1, float(self.sigma_data), float(self.sigma_min), _p(sc[0]), _p(sc[1]), B, stream),
              "cm scalars")
        eng = self.net._engine(B, T, dev, lane)
        return eng.forward(sample, sigma, cond, in_scale=None, c_out=sc[0], c_skip=sc[1], skip_src=sample, train=train,
                           dropout_seed=dropout_seed, infer=infer)

    def forward(self, sample, sigma, cond_sample=None, cond=None, _check_range=True):
        """consistency_model.py:63-79."""
        engine.require_device(sample)
        if cond_sample is not None:
            raise NotImplementedError("cond_sample concatenation is not used by any 1-D consistency config")
        sample, sigma = sample.contiguous(), sigma.contiguous().float()
        B = sample.shape[0]
        lanes = 1  # one forward cannot amortise 4x the launches (measured 8.3 vs 6.5 ms at B = 64); kept for TQDNE experiments
        if os.environ.get("TQDNE_CM_LANES"):
            lanes = int(os.environ["TQDNE_CM_LANES"])
        if lanes > 1 and (B % lanes or torch.is_grad_enabled()):
            lanes = 1
        if lanes < 2:
            infer = not torch.is_grad_enabled()
            y = self._forward_static(sample, sigma, cond, infer=infer).clone()
            if infer and _check_range and self.net._engine(B, sample.shape[2], sample.device, 0).check_range():
                y = self._forward_static(sample, sigma, cond, infer=infer).clone()  # (the plan is on bf16x3 now)
            return y
        # independent samples: sub-batches on separate HIP streams run out of phase (see LightningEDM.sample_deterministically)
        dev = sample.device
        h = B // lanes
        main = torch.cuda.current_stream(dev)
        out = torch.empty_like(sample[:, : self.net.out_channels])
        for i in range(lanes):
            st = main if i == 0 else self._side_stream(dev, i)
            if i:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                sl = slice(i * h, (i + 1) * h)
                y = self._forward_static(sample[sl].contiguous(), sigma[sl].contiguous(),
                                         None if cond is None else cond[sl].contiguous(), lane=engine.CONCURRENT_LANE0 + i, infer=True)
                out[sl].copy_(y)
        for i in range(1, lanes):
            main.wait_stream(self._side_stream(dev, i))
        return out

    def _side_stream(self, dev, i):
        return engine.side_stream(dev, i)   # (one pool per process, see engine.side_stream)

    @torch.no_grad()
    def sample(self, shape, sigmas=[1.0], cond_sample=None, cond=None):
        """consistency_model.py:81-106 (the refinement noise is uniform, ``rand_like``, as in the reference)."""
        epsilon = torch.randn(shape, device=self.device)
        return self.sample_from(epsilon, sigmas, [torch.rand_like(epsilon) for _ in sigmas], cond_sample, cond)

    @torch.no_grad()
    def sample_from(self, epsilon, sigmas, uniform_noises, cond_sample=None, cond=None):
        def run():
            ones = torch.ones(epsilon.shape[0], device=epsilon.device)
            sample = self(epsilon, ones * self.sigma_max, cond_sample, cond, _check_range=False)
            for sigma, u in zip(sigmas, uniform_noises):
                sample = sample + u * sigma
                sample = self(sample, ones * sigma, cond_sample, cond, _check_range=False)
            return sample
        out = run()
        # range guard of the fp16-range conv scheme: ONE flag read per sample call (as the EDM samplers do), not one per network
        # evaluation; if a tensor came near the fp16 range the plans are on bf16x3 now and the sampling is repeated
        eng = self.net._engine(epsilon.shape[0], epsilon.shape[2], epsilon.device, 0)
        if eng.check_range():
            out = run()
        return out

    # ------------------------------------------------------------------ iCT training (consistency_model.py:115-190)
    def _schedule(self):
        """consistency_model.py:121-138.  ``trainer.max_steps`` / ``global_step`` are Lightning's; without a trainer the
        attributes ``max_steps`` / ``global_step`` of the module are used."""
        tr = getattr(self, "trainer", None)
        max_steps = tr.max_steps if tr is not None else getattr(self, "max_steps", 1)
        global_step = getattr(self, "global_step", 0)
        prime = np.floor(max_steps / (np.log2(np.floor(self.final_timesteps / self.initial_timesteps)) + 1))
        num = self.initial_timesteps * 2 ** np.floor(global_step / prime)
        num = min(num, self.final_timesteps) + 1
        rho_inv = 1.0 / self.rho
        steps = torch.arange(num, device=self.device) / (num - 1)
        sig = self.sigma_min**rho_inv + steps * (self.sigma_max**rho_inv - self.sigma_min**rho_inv)
        return sig**self.rho

    def step(self, batch):
        """A single step of training or validation (consistency_model.py:115-176): teacher at sigma_t (no gradient, same
        dropout masks as the student), student at sigma_{t+1}, weighted pseudo-Huber distance.  Both UNet passes and the
        backward are HIP; the schedule, the (B,)-sized draws and the loss on the (B, C, T) outputs are torch glue."""
        sample = batch["signal"]
        if "cond_signal" in batch:
            raise NotImplementedError("cond_signal is not used by any 1-D consistency config")
        cond = batch["cond"] if "cond" in batch else None
        sigmas = self._schedule()
        z = lambda s_: torch.erf((torch.log(s_) - self.lognormal_mean) / (self.lognormal_std * np.sqrt(2)))
        pdf = z(sigmas[1:]) - z(sigmas[:-1])
        pdf = pdf / pdf.sum()
        timesteps = torch.multinomial(pdf, sample.shape[0], replacement=True)
        epsilon = torch.randn_like(sample)
        return _ICTLossFn.apply(self, sample.contiguous(), sigmas, timesteps, epsilon, cond, *self.net.parameters())

    def training_step(self, batch, batch_idx: int):
        loss = self.step(batch)
        self.log("train_loss", loss.item(), prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx: int):
        loss = self.step(batch)
        self.log("val_loss", loss.item())
        return loss

    def configure_optimizers(self):
        return torch.optim.RAdam(self.net.parameters(), lr=self.lr)

    def evaluate(self, batch, sigmas=[1]):
        sample = batch["signal"]
        cond_sample = batch["cond_signal"] if "cond_signal" in batch else None
        cond = batch["cond"] if "cond" in batch else None
        return self.sample(sample.shape, sigmas, cond_sample, cond)
