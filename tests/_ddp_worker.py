"""Worker of tests/test_ddp_gpu.py (one process per rank; launched with RANK / WORLD_SIZE / MASTER_* / TQ_TEST_BACKEND set).

Real model, real kernels: every rank trains the micro EDM UNet on its shard of a fixed global batch with injected noise;
after the exchange the gradients must equal the one-rank full-batch gradients, with and without
the overlap of the exchange with the backward sweep, and after the optimizer step all replicas must hold identical weights.
With fewer GPUs than ranks (the 1-GPU test box) the ranks share cuda:0 and the exchange goes through gloo, staged through
host memory by the test's trainer subclass (the product path is RCCL: backend "nccl" when every rank has its own GPU)."""

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Done:
    def wait(self):
        return True


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TQ_TEST_BACKEND", "gloo")
    ngpu = torch.cuda.device_count()
    dev = torch.device("cuda", rank % max(ngpu, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        from tqdne_amd.trainer import init_process_group   # (side streams first, then the communicator)
        init_process_group("nccl", device=dev, rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from conftest import cfg_of, load_golden, rel_err
    from tqdne_amd import LightningEDM, rng
    from tqdne_amd.autograd import edm_loss_and_grads
    from tqdne_amd.trainer import DataParallelTrainer, shard_batch

    class Trainer(DataParallelTrainer):
        def _allreduce_async(self, t):
            if backend == "nccl":
                return super()._allreduce_async(t)
            h = t.cpu()  # (synchronises: the slice is final on the stream at this point)
            dist.all_reduce(h)
            t.copy_(h)
            return _Done()

    which = os.environ.get("TQ_TEST_CONFIG", "micro")
    if which == "paper":   # the configuration of BASELINE cfg2 (the paper UNet under data parallelism), at a length the test can afford
        from test_hip_unet import perturbed_state
        from tqdne_amd import UNetModel, paper_1d_unet_config
        cfg = dict(paper_1d_unet_config(), dropout=0.0)
        torch.manual_seed(0)
        sd = perturbed_state(UNetModel(**cfg), 31)
        Bg, T = 4, 512
    else:
        sd, d = load_golden("micro_unet.npz")
        cfg = dict(cfg_of(d), dropout=0.0)  # masks are indexed by the position in the LOCAL batch: equivalence needs p = 0
        Bg, T = 8, 256
    g = torch.Generator().manual_seed(11)
    batch = {"signal": 0.5 * torch.randn(Bg, 3, T, generator=g), "cond": torch.randn(Bg, 5, generator=g)}
    eps_g, noise_g = torch.randn(Bg, generator=g), torch.randn(Bg, 3, T, generator=g)

    class InjectedEDM(LightningEDM):
        """step_and_backward with the two random draws of edm.py:126,128 taken from the fixed global draws"""
        inject = None

        def step_and_backward(self, batch, on_bucket=None, bucket_elems=4 << 20):
            eps, noise = self.inject
            return edm_loss_and_grads(self, batch["signal"].contiguous(), eps, noise, batch.get("cond"), on_bucket=on_bucket,
                                      bucket_elems=bucket_elems)

    def make():
        m = InjectedEDM(cfg, {"learning_rate": 1e-3, "max_steps": 10, "eta_min": 0.0})
        m.unet.load_state_dict(sd)
        return m.to(dev).train()

    rng.seed_rank(0, rank)
    res = {}
    # reference: the full global batch on this rank alone
    full = make()
    full.inject = (eps_g.to(dev), noise_g.to(dev))
    loss_full, flat_full = full.step_and_backward({k: v.to(dev) for k, v in batch.items()})
    n_grad = full.unet._engine(Bg, T, dev)._bwd.n_grad
    g_full = flat_full[:n_grad].clone()
    offs_full = full.unet._engine(Bg, T, dev)._bwd.offs

    for overlap in (True, False):
        m = make()
        local = {k: v.to(dev) for k, v in shard_batch(batch, rank, world).items()}
        per = Bg // world
        m.inject = (eps_g[rank * per:(rank + 1) * per].to(dev), noise_g[rank * per:(rank + 1) * per].contiguous().to(dev))
        tr = Trainer(m, world_size=world, bucket_bytes=(16 << 20) if which == "paper" else (64 << 10), overlap=overlap, fused_optimizer=True)
        # exchange only (optimizer held back): capture the reduced gradients
        opt_step = tr.optimizer.step
        grabbed = {}

        def hold(grad_scale=1.0, skip_flag=None, _tr=tr, _m=m):
            bwd = _m.unet._engine(per, T, dev)._bwd
            grabbed["g"] = (bwd.flat[:bwd.n_grad] * grad_scale).clone()
            grabbed["offs"] = dict((id(p), bwd.offs[id(p)]) for p in _m.unet.parameters())
            return opt_step(grad_scale=grad_scale)

        tr.optimizer.step = hold
        loss = tr.train_step(local)
        torch.cuda.synchronize()
        # the two plans (B = 8 and B = 4) lay their gradients out identically (the layout depends on the model only)
        assert [grabbed["offs"][id(p)] for p in m.unet.parameters()] == [offs_full[id(p)] for p in full.unet.parameters()]
        err = rel_err(grabbed["g"].cpu(), g_full.cpu())
        # per tensor, relative to that tensor's own scale; tensors whose true gradient is zero (a conv bias in front of a
        # one-channel-per-group GroupNorm: the micro net has 32 channels) hold rounding noise only, hence the floor
        worst, floor = 0.0, 1e-4 * float(g_full.abs().max())
        for p_l, p_f in zip(m.unet.parameters(), full.unet.parameters()):
            if not p_l.requires_grad:
                continue
            o = offs_full[id(p_f)]
            a, b = grabbed["g"][o:o + p_l.numel()], g_full[o:o + p_l.numel()]
            worst = max(worst, float((a - b).abs().max()) / max(float(b.abs().max()), floor))
        # replicas identical after the update
        chk = torch.cat([p.detach().reshape(-1) for p in m.unet.parameters()]).double().cpu()
        sums = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(sums, torch.stack([chk.sum(), (chk * chk).sum()]))
        losses = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(losses, torch.tensor([float(loss)], dtype=torch.float64))
        res["overlap" if overlap else "after"] = dict(
            err_flat=err, err_worst_tensor=worst, replicas_equal=bool(all(torch.equal(s, sums[0]) for s in sums)),
            buckets=list(tr.last_bucket_sizes), loss_mean=float(sum(l.item() for l in losses) / world), loss_full=float(loss_full))
    if rank == 0:
        print("DDP_RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
