"""DDPM training / ancestral sampling: drop-in for ``tqdne.diffusion.LightningDDMP`` (reference tqdne/diffusion.py:9-128: forward
55-65, sample 67-79, evaluate 81-86, step 88-109, configure_optimizers 117-128) on the HIP UNet path.

PARITY UNPINNED.  The reference takes its scheduler from ``diffusers`` (``DDPMScheduler``) and its learning-rate schedule from
``diffusers.optimization`` -- a dependency that is absent from the reference's own lockfile (the module cannot be imported in the
reference's environment) and from this image.  ``DDPMScheduler`` below restates the *published* algorithm (Ho, Jain, Abbeel 2020,
"Denoising Diffusion Probabilistic Models", eq. 4, 7, 11 and section 3.2/3.4) with the defaults diffusers documents for the class
(1000 training steps, linear betas 1e-4 .. 0.02, fixed-small variance, x0 clipped to [-1, 1], "leading" inference spacing); it is
checked against ``oracle/diffusion.py`` (an independent float64 restatement of the same equations) and through algebraic properties
(tests/test_diffusion.py), not against diffusers' output.  What IS the reference's own code -- the module surface, the channel
concat of the conditioning signal, the epsilon / sample target, the MSE loss, AdamW -- follows tqdne/diffusion.py line by line.
The network is a ``tqdne_amd.UNetModel``: forward and backward run on the HIP plan, the scheduler's elementwise arithmetic in
``tq_scale_add2`` / ``tq_ddpm_step``.
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from . import _lib, engine, rng
from ._lib import check
from .lightning_compat import LightningModule


def _p(t):
    return None if t is None else t.data_ptr()


class DDPMScheduler:
    """The part of diffusers' ``DDPMScheduler`` that tqdne/diffusion.py touches: ``config.num_train_timesteps``, ``timesteps``,
    ``set_timesteps``, ``add_noise`` (line 98), ``step(...).prev_sample`` (line 77)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", prediction_type: str = "epsilon", clip_sample: bool = True,
                 clip_sample_range: float = 1.0, variance_type: str = "fixed_small", trained_betas=None,
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0,
                 timestep_spacing: str = "leading", steps_offset: int = 0, rescale_betas_zero_snr: bool = False):
        # the remaining constructor arguments of the diffusers class are accepted at their documented defaults only: anything else
        # would select arithmetic this restatement does not have (and cannot pin: diffusers is not in the reference's lockfile)
        for name, val, default in (("trained_betas", trained_betas, None), ("thresholding", thresholding, False),
                                   ("timestep_spacing", timestep_spacing, "leading"), ("steps_offset", steps_offset, 0),
                                   ("rescale_betas_zero_snr", rescale_betas_zero_snr, False)):
            if val != default:
                raise NotImplementedError(f"DDPMScheduler({name}={val!r}): only the default ({default!r}) is restated")
        if beta_schedule != "linear":
            raise NotImplementedError("only the linear beta schedule (the class default) is restated")
        if prediction_type not in ("epsilon", "sample"):
            raise NotImplementedError("prediction_type must be 'epsilon' or 'sample' (what LightningDDMP accepts)")
        if variance_type not in ("fixed_small", "fixed_large"):
            raise NotImplementedError("variance_type: fixed_small (default) or fixed_large")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range, variance_type=variance_type)
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)            # abar_t  (Ho et al., eq. 4)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self._dev = {}

    def set_timesteps(self, num_inference_steps: int):
        """"leading" spacing: multiples of T // n, descending."""
        T = self.config.num_train_timesteps
        if not 0 < num_inference_steps <= T:
            raise ValueError("num_inference_steps must be in (0, num_train_timesteps]")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).flip(0)

    # ---- forward process q(x_t | x_0) = N(sqrt(abar_t) x_0, (1 - abar_t) I)   (eq. 4)
    def _tables(self, device):
        key = str(device)
        if key not in self._dev:
            ac = self.alphas_cumprod.to(device)
            self._dev[key] = (ac.sqrt().contiguous(), (1.0 - ac).sqrt().contiguous())
        return self._dev[key]

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        engine.require_device(original_samples)
        sa, sb = self._tables(original_samples.device)
        t = timesteps.to(original_samples.device).long()
        a, c = sa[t].contiguous(), sb[t].contiguous()      # (B,) lookups: indexing glue
        x, n = original_samples.contiguous().float(), noise.contiguous().float()
        out = torch.empty_like(x)
        B = x.shape[0]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        check(_lib.load().tq_scale_add2(_p(x), _p(n), _p(a), _p(c), _p(out), B, x[0].numel(), stream), "ddpm add_noise")
        return out

    # ---- reverse process, one ancestral step (eq. 7 posterior mean with x0 from eq. 15 / the network, section 3.2 variance)
    def step_coefficients(self, t: int):
        """host scalars of one step, in float64: (sqrt(1 - abar_t), 1 / sqrt(abar_t), coef_x0, coef_xt, sigma)."""
        T = self.config.num_train_timesteps
        n = self.num_inference_steps or T
        prev_t = t - T // n
        ac = self.alphas_cumprod.double()
        abar_t = float(ac[t])
        abar_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
        beta_prod_t, beta_prod_prev = 1.0 - abar_t, 1.0 - abar_prev
        cur_alpha = abar_t / abar_prev
        cur_beta = 1.0 - cur_alpha
        coef_x0 = math.sqrt(abar_prev) * cur_beta / beta_prod_t
        coef_xt = math.sqrt(cur_alpha) * beta_prod_prev / beta_prod_t
        if t > 0:
            var = cur_beta if self.config.variance_type == "fixed_large" else max(beta_prod_prev / beta_prod_t * cur_beta, 1e-20)
            sigma = math.sqrt(var)
        else:
            sigma = 0.0
        return math.sqrt(beta_prod_t), 1.0 / math.sqrt(abar_t), coef_x0, coef_xt, sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, noise: torch.Tensor = None, generator=None,
             return_dict: bool = True):
        """``noise``: the step's N(0, I) draw (tests inject it); default: drawn here -- from ``generator`` when given -- as the
        reference's scheduler does."""
        engine.require_device(sample)
        t = int(timestep)
        s1, inv, c0, ct, sigma = self.step_coefficients(t)
        x, mo = sample.contiguous().float(), model_output.contiguous().float()
        if sigma > 0.0 and noise is None:
            noise = torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=generator)
        z = noise.contiguous().float() if (sigma > 0.0 and noise is not None) else None
        out = torch.empty_like(x)
        clip = float(self.config.clip_sample_range) if self.config.clip_sample else 0.0
        stream = torch.cuda.current_stream(x.device).cuda_stream
        check(_lib.load().tq_ddpm_step(_p(x), _p(mo), _p(z), _p(out), x.numel(), int(self.config.prediction_type == "epsilon"),
                                       s1, inv, clip, c0, ct, sigma, stream), "ddpm step")
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
    """Linear warm-up to the base rate, then half a cosine down to zero -- the schedule diffusion.py:120-127 asks diffusers for."""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


class _DDPMLossFn(torch.autograd.Function):
    """loss = mean((net(cat(cond_signal, x_t), t) - target)^2) with x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise (diffusion.py:88-109):
    HIP forward in the module's mode, HIP backward seeded with d loss / d pred (parameter gradients only)."""

    @staticmethod
    def forward(ctx, module, signal, cond_signal, cond, noise, timesteps, *params):
        lib = _lib.load()
        dev = signal.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        noisy = module.noise_scheduler.add_noise(signal, noise, timesteps)
        x_in = torch.cat((cond_signal.float(), noisy), dim=1).contiguous() if module.cond_signal_input else noisy
        B, _, T = x_in.shape
        eng = module.net._engine(B, T, dev)
        train = module.training
        # (infer=False: the launches keep what a backward reads, in train and in eval mode; dropout only in train mode)
        pred = eng.forward(x_in, timesteps.float(), cond if module.cond_input else None, train=train,
                           dropout_seed=rng.next_dropout_seed() if train else 0)
        target = (noise if module.prediction_type == "epsilon" else signal).contiguous().float()
        loss = torch.empty(1, device=dev)
        dpred = torch.empty_like(pred)
        check(lib.tq_mse_loss(_p(pred), _p(target), _p(loss), _p(dpred), pred.numel(), stream), "mse loss")
        ctx.eng, ctx.dpred, ctx.fwd_id, ctx.train = eng, dpred, eng._fwd_count, train
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.eng
        if eng._fwd_count != ctx.fwd_id:
            raise RuntimeError("another forward of the same shape ran between this forward and its backward")
        grads = eng.backward(ctx.dpred, gloss)
        return (None,) * 6 + tuple(grads)


class LightningDDMP(LightningModule):
    """A Lightning module for training a diffusion model (tqdne/diffusion.py:9-53; same constructor arguments)."""

    def __init__(self, net: torch.nn.Module, noise_scheduler: DDPMScheduler, optimizer_params: dict,
                 prediction_type: str = "epsilon", cond_signal_input: bool = False, cond_input: bool = False):
        super().__init__()
        self.net = net
        self.optimizer_params = optimizer_params
        self.noise_scheduler = noise_scheduler
        if prediction_type not in ["epsilon", "sample"]:
            raise ValueError(f"Unknown prediction type {prediction_type}")
        self.prediction_type = prediction_type
        self.cond_signal_input = cond_signal_input
        self.cond_input = cond_input
        self.save_hyperparameters()

    def log_value(self, value, name, train=True, prog_bar=True):
        if train:
            self.log(f"train_{name}", value, prog_bar=prog_bar)
        else:
            self.log(f"val_{name}", value, prog_bar=prog_bar)

    def forward(self, input, t, cond_signal=None, cond=None):
        """Make a forward pass through the network (diffusion.py:55-65)."""
        if self.cond_signal_input:
            assert cond_signal is not None
            input = torch.cat((cond_signal, input), dim=1)
        cond = cond if self.cond_input else None
        return self.net(input.contiguous(), t, cond=cond)

    @torch.no_grad()
    def sample(self, shape, cond_signal=None, cond=None, start=None, noises=None):
        """Sample from the diffusion model (diffusion.py:67-79).  ``start`` / ``noises`` (one tensor per step): the random draws,
        injected by tests; default: drawn here like the reference."""
        sample = torch.randn(shape, device=self.device) if start is None else start
        for i, t in enumerate(self.noise_scheduler.timesteps):
            pred = self.forward(sample, t * torch.ones(shape[0], device=self.device), cond_signal, cond)
            sample = self.noise_scheduler.step(pred, t, sample, noise=None if noises is None else noises[i]).prev_sample
        return sample

    def evaluate(self, batch):
        """Evaluate diffusion model (diffusion.py:81-86)."""
        shape = batch["signal"].shape
        cond_signal = batch["cond_signal"] if self.cond_signal_input else None
        cond = batch["cond"] if self.cond_input else None
        return self.sample(shape, cond_signal, cond)

    def step(self, batch, train):
        """diffusion.py:88-109."""
        signal_batch = batch["signal"]
        noise = torch.randn(signal_batch.shape, device=signal_batch.device)
        timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (signal_batch.shape[0],),
                                  device=signal_batch.device).long()
        loss = self.step_with_noise(batch, noise, timesteps)
        self.log_value(loss, "loss", train=train, prog_bar=True)
        return loss

    def step_with_noise(self, batch, noise, timesteps):
        """``step`` with the two random draws of diffusion.py:93-97 supplied by the caller."""
        signal = batch["signal"]
        engine.require_device(signal)
        cond_signal = batch["cond_signal"] if self.cond_signal_input else None
        cond = batch["cond"] if self.cond_input else None
        params = list(self.net.parameters())
        return _DDPMLossFn.apply(self, signal.contiguous().float(), cond_signal, cond, noise, timesteps, *params)

    def training_step(self, batch, batch_idx: int):
        return self.step(batch, train=True)

    def validation_step(self, batch, batch_idx: int):
        return self.step(batch, train=False)

    def configure_optimizers(self):
        optimizer = torch.optim.AdamW(self.net.parameters(), lr=self.optimizer_params["learning_rate"])
        lr_scheduler = get_cosine_schedule_with_warmup(
            optimizer=optimizer, num_warmup_steps=self.optimizer_params["lr_warmup_steps"],
            num_training_steps=(self.optimizer_params["n_train"] * self.optimizer_params["max_epochs"]))
        return [optimizer], [lr_scheduler]
