"""Data-parallel training loop for the EDM module: stands where Lightning's fit loop + DDPStrategy stand in the
reference (experiments/train_1d_edm.py:34-70): one process per GPU, identical replicas, the batch is
sharded by rank, gradients are summed with RCCL all-reduce over xGMI (torch.distributed backend "nccl") and divided
by the world size, then every rank applies the same Adam + per-step cosine LR update (edm.py:240-251).

The exchange is overlapped with the backward, the way torch's DDP reducer does it for the reference: the backward plan lays
the flat gradient buffer out in the order its reverse sweep finalises the gradients (head, output blocks, middle block,
input blocks, stem, embedding MLPs) and calls back as each contiguous bucket becomes final; the bucket's all-reduce is
issued right there (asynchronously: on RCCL's own stream, ordered behind the launches enqueued so far) and runs under the
rest of the sweep.  Only the optimizer launch waits for the exchange.  Every rank cuts and issues the buckets identically
(the cut depends on the model and ``bucket_bytes`` only), so the collectives match up by construction.
"""

from __future__ import annotations

import inspect

import torch
import torch.distributed as dist


def init_process_group(backend: str = "nccl", device=None, **kw):
    """``torch.distributed.init_process_group`` with the one ordering rule this package has on ROCm: the sampler lanes and the
    backward's weight-gradient stream make their first submission BEFORE the RCCL communicator creates its streams.  ROCm binds a
    HIP stream to a hardware queue at its first submission; with the communicator in first, the backward's two streams share a
    queue and the training step of the paper UNet runs 22.1 -> 28.1 ms (measured with a forced one-rank exchange,
    ``profiles/r04_u_rccl_ab.txt``).  Extra keywords go to torch (``rank``, ``world_size``, ``timeout``, ``device_id`` ...)."""
    if device is not None and torch.device(device).type == "cuda":
        from .engine import reserve_side_streams
        reserve_side_streams(torch.device(device))
    return dist.init_process_group(backend, **kw)


def allreduce_mean_(flat, world_size: int, bucket_elems: int = 8 << 20, scale: bool = True):
    """In-place mean of ``flat`` over all ranks: a few large bucketed all-reduces (RCCL over xGMI on GPUs, gloo in the
    CPU tests), launched asynchronously and waited together, then one scale.  62 MB of gradients = 2 buckets of 32 MB."""
    if world_size <= 1:
        return flat
    flats = list(flat) if isinstance(flat, (list, tuple)) else [flat]  # (the autoencoder step returns per-tensor gradients)
    works = []
    for f in flats:
        f1 = f.view(-1)
        works += [dist.all_reduce(f1[i:i + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                  for i in range(0, f1.numel(), bucket_elems)]
    for w in works:
        w.wait()
    if scale:
        for f in flats:
            f.mul_(1.0 / world_size)
    return flat


def shard_batch(batch: dict, rank: int, world_size: int) -> dict:
    """Rank ``rank``'s slice of a global batch (what Lightning's DistributedSampler does for the reference)."""
    if world_size <= 1:
        return batch
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        assert n % world_size == 0, "global batch must divide evenly over the ranks"
        per = n // world_size
        out[k] = v[rank * per:(rank + 1) * per]
    return out


class _Done:
    """handle of a collective that has already completed (host-staged gloo exchange)"""

    def wait(self):
        return True


class DataParallelTrainer:
    def __init__(self, module, world_size: int = 1, bucket_bytes: int = 16 << 20, ema_decay=None, fused_optimizer=None,
                 overlap: bool = True, process_group=None, force_exchange: bool = False):
        """``force_exchange``: issue the collectives with ONE rank too (a sum over one rank is the identity: the self-test of the
        RCCL path on a 1-GPU box -- communicator, RCCL's stream behind the sweep's launches, the waits in front of the optimizer).
        ``ema_decay``: keep the EMA weights of the reference's EMA callback (tqdne/ema.py; 0.999 in the reference's runs).
        ``fused_optimizer``: one-launch Adam (+ EMA) instead of torch.optim.Adam; default: on GPUs.
        ``overlap``: issue each gradient bucket's all-reduce from inside the backward sweep (default) instead of after it.
        ``bucket_bytes``: 16 MB = 4 buckets over the paper UNet's 62 MB of gradients (xGMI rings are per-link bound: few,
        large messages; the first bucket leaves after the output blocks' half of the sweep)."""
        self.module = module
        self.world = world_size
        self.group = process_group
        self.overlap = overlap
        self.exchange = world_size > 1 or force_exchange
        cfg = module.configure_optimizers()
        self.optimizer = cfg["optimizer"]
        self.scheduler = cfg["lr_scheduler"]["scheduler"]
        on_gpu = next(module.parameters()).device.type == "cuda"
        self.fused = on_gpu if fused_optimizer is None else fused_optimizer
        self.ema_decay = ema_decay
        self._ema = None
        if self.fused:
            from .optim import FusedAdamEMA
            g = self.optimizer.param_groups[0]
            wd = g.get("weight_decay", 0.0) if isinstance(self.optimizer, torch.optim.AdamW) else 0.0
            self.optimizer = FusedAdamEMA(module.named_parameters(), lr=g["lr"], betas=g["betas"], eps=g["eps"],
                                          ema_decay=ema_decay, weight_decay=wd)
            op = module.optimizer_params
            self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=op["max_steps"],
                                                                        eta_min=op["eta_min"])
        elif ema_decay is not None:
            self._ema = {n: p.detach().clone() for n, p in module.named_parameters() if p.requires_grad}
        self.bucket_elems = max(1, bucket_bytes // 4)
        sig = inspect.signature(module.step_and_backward).parameters
        self._hooked = "on_bucket" in sig
        self._tailed = "tail_fill" in sig   # the module lets a few words of ours ride at the end of the last gradient bucket
        self.last_bucket_sizes = []   # elements of each bucket exchanged by the last step, in issue order (diagnostics / tests)
        if self.exchange:
            self._broadcast_parameters()

    # ------------------------------------------------------------------ exchange
    def _broadcast_parameters(self):
        """Replicas start identical (DDP's initial broadcast): rank 0's parameters go out as ONE flat buffer per dtype
        (round 4 issued one collective per tensor: 311 for the paper UNet) and are copied back in place on the parameters themselves
        (not p.data), so that their version counters move and the packed weights follow."""
        with torch.no_grad():
            by_dtype = {}
            for p in self.module.parameters():
                by_dtype.setdefault(p.dtype, []).append(p)
            for ps in by_dtype.values():
                flat = torch.cat([p.detach().reshape(-1) for p in ps])
                if self._host_staged(flat):
                    h = flat.cpu()
                    dist.broadcast(h, src=0, group=self.group)
                    flat.copy_(h)
                else:
                    dist.broadcast(flat, src=0, group=self.group)
                o = 0
                views = []
                for p in ps:
                    views.append(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()
                torch._foreach_copy_(ps, views)

    def _host_staged(self, t: torch.Tensor) -> bool:
        """device tensors over the gloo backend (the dry-run configuration of bench.py / the 1-GPU tests: ranks sharing a device) go
        through host memory: the product backend for GPUs is "nccl" = RCCL, which takes device tensors as they are"""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _allreduce_async(self, t: torch.Tensor):
        """start the sum all-reduce of ``t`` (in place); returns a handle with ``wait()``.  NCCL/RCCL: the collective is
        enqueued on the process group's own stream behind everything the current stream holds so far."""
        if self._host_staged(t):
            h = t.cpu()   # (synchronises: the slice is final on the current stream at this point)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return _Done()
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True, group=self.group)

    def train_step(self, batch):
        """loss, backward (HIP) with the gradient exchange riding under it, Adam, cosine LR.  Returns the local loss (a device
        scalar; no host sync here -- the reference's ``loss.item()`` logging, edm.py:138, belongs to the caller)."""
        works = []
        self.last_bucket_sizes = []
        self.last_tail_words = 0      # words of ours that rode at the end of the last gradient bucket (the range-guard pair: 2)
        hook = None
        if self.exchange and self.overlap and self._hooked:
            def hook(sl):
                self.last_bucket_sizes.append(sl.numel())
                works.append(self._allreduce_async(sl))
        tail = None
        if hook is not None and self._tailed and self._skip_exchange_live():
            # the range-guard pair of this step rides at the end of the last gradient bucket (no collective of its own)
            tail = []

            def tail_fill(words):
                self._skip_fill(words)
                tail.append(words)
                self.last_tail_words = words.numel()
        if tail is not None:
            loss, flat = self.module.step_and_backward(batch, on_bucket=hook, bucket_elems=self.bucket_elems, tail_fill=tail_fill)
        elif self._hooked:
            loss, flat = self.module.step_and_backward(batch, on_bucket=hook, bucket_elems=self.bucket_elems)
        else:
            loss, flat = self.module.step_and_backward(batch)
        if self.exchange and hook is None:
            flats = list(flat) if isinstance(flat, (list, tuple)) else [flat]
            for f in flats:
                f1 = f.view(-1)
                for i in range(0, f1.numel(), self.bucket_elems):
                    self.last_bucket_sizes.append(min(self.bucket_elems, f1.numel() - i))
                    works.append(self._allreduce_async(f1[i:i + self.bucket_elems]))
        reduced = tail[0] if tail else None
        if reduced is None:
            skip = self._range_skip_flag()
        for w in works:   # (NCCL: makes the current stream wait for the exchange; the host does not block)
            w.wait()
        if reduced is not None:
            skip = self._range_skip_flag(reduced)   # (after the waits: every AGREE_EVERY steps the host reads the reduced pair)
        self.last_skip = skip
        if self.fused:
            # the 1 / world_size of the gradient mean rides in the optimizer launch
            self.optimizer.step(grad_scale=1.0 / self.world, skip_flag=skip)
        else:
            if self.world > 1:
                flats = list(flat) if isinstance(flat, (list, tuple)) else [flat]
                for f in flats:
                    f.mul_(1.0 / self.world)
            self.optimizer.step()
            if self._ema is not None:
                with torch.no_grad():
                    named = dict(self.module.named_parameters())
                    torch._foreach_lerp_(tuple(self._ema.values()), tuple(named[n] for n in self._ema), 1 - self.ema_decay)
        self.scheduler.step()
        return loss

    AGREE_EVERY = 64   # steps between the (host-synchronous) checks whether every rank has left the fp16-range scheme

    def _local_range_flag(self):
        """(device int32[1] range-guard flag of this rank's model | None, is this rank still on the fp16-range scheme?)"""
        unet = getattr(self.module, "unet", None)
        if not self.fused or unet is None or getattr(unet, "dims", 1) != 1:
            return None, False
        dev = next(self.module.parameters()).device
        from .engine import shared_range_flag
        return shared_range_flag(unet, dev), getattr(unet, "_conv_scheme", "auto") == "auto"

    def _range_skip_flag(self, reduced=None):
        """Device predicate of the optimizer launch (fused optimizer only): the range-guard flag of the fp16-range forward scheme.
        The host learns of a raised flag one or more steps late (engine._range_poll does not synchronise), so the step whose
        forward raised it -- activations within a factor two of the fp16 range, possibly inf / NaN gradients -- is dropped ON
        THE DEVICE, and so is every later step until the host has moved the plans to bf16x3.  With several ranks the flag is
        summed over the ranks first (a rank that skipped alone would leave the replicas different); a rank already on bf16x3
        contributes 0 (its kernels still raise the flag for large activations, which that scheme handles).

        Bookkeeping of a dropped step: only the device-side update is skipped.  The host-side counters move on -- Adam's step count
        (bias corrections), the cosine schedule, the parameters' version counters (one redundant re-pack) -- exactly as if the step had
        produced a zero update; at most a handful of steps per run are affected (the host switches the plans to bf16x3 within a
        step or two of the flag).

        The exchange ends by agreement: the second word of the exchanged pair counts the ranks that are off the fp16-range scheme;
        every ``AGREE_EVERY`` steps all ranks read it (the same step on every rank, so they stop together) and, once it equals the
        world size, no rank issues the collective any more.

        ``reduced`` (round 5): the pair already summed over the ranks -- it rode at the end of the step's last gradient bucket
        (``tail_fill`` of BackwardPlan.run), so the step has no collective of its own for it; None: the pair is exchanged here (modules
        without the bucket hook, ``overlap=False``)."""
        flag, auto = self._local_range_flag()
        if flag is None:
            return None
        if not self.exchange:
            return flag if auto else None
        if getattr(self, "_skip_done", False):
            return None
        if reduced is None:
            # (summed as floats through the same exchange path as the gradients; the kernel tests word 0 for "non-zero", and a sum
            # of 0 / 1 flags has a non-zero bit pattern exactly when some rank raised its flag)
            if getattr(self, "_skip", None) is None:
                self._skip = torch.zeros(2, dtype=torch.float32, device=flag.device)
            self._skip_fill(self._skip)
            self._allreduce_async(self._skip).wait()
            reduced = self._skip
        self._skip_steps = getattr(self, "_skip_steps", 0) + 1
        if self._skip_steps % self.AGREE_EVERY == 0 and float(reduced[1].item()) >= self.world:
            self._skip_done = True
        return reduced

    def _skip_exchange_live(self) -> bool:
        """does this step exchange the range-guard pair? (fused optimizer on a 1-D UNet, exchange on, no agreement reached yet)"""
        flag, _ = self._local_range_flag()
        return flag is not None and self.exchange and not getattr(self, "_skip_done", False)

    def _skip_fill(self, words):
        """this rank's pair into ``words`` (2 floats): [its range-guard flag while it is on the fp16-range scheme, 1 once it left it]"""
        flag, auto = self._local_range_flag()
        if auto:
            words[0:1].copy_(flag)
            words[1:2].zero_()
        else:
            words[0:1].zero_()
            words[1:2].fill_(1.0)

    def ema_state(self):
        """name -> EMA weights, as the reference's EMA callback stores them in a checkpoint (tqdne/ema.py:50-51)."""
        if self.fused:
            return self.optimizer.ema_state()
        if self._ema is None:
            raise RuntimeError("constructed without ema_decay")
        return self._ema
