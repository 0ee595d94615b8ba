"""Worker of tests/test_torch_ddp_gpu.py (one process per rank; RANK / WORLD_SIZE / MASTER_* set by the test).

The route the REFERENCE takes to several GPUs (experiments/train_1d_edm.py:34-41,65-70): Lightning wraps the LightningModule in
``torch.nn.parallel.DistributedDataParallel``, redirects the wrapper's ``forward`` to ``training_step`` and calls
``loss.backward()``; the DDP reducer's autograd hooks bucket and all-reduce ``p.grad`` while the backward runs, then
``torch.optim.Adam.step()``.  Here ``loss.backward()`` is ONE custom autograd Function (tqdne_amd/autograd.py) that hands back all 310
gradients at once from the HIP backward plan -- this worker checks that the reducer copes with that: after the backward every rank's
``p.grad`` equals the one-rank full-batch gradient, and after the optimizer steps the replicas hold identical weights that match the
one-rank run's.  With fewer GPUs than ranks the ranks share cuda:0 and the process group is gloo (CUDA tensors staged by gloo)."""

import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def redirected_step(ddp, module, *args, **kw):
    """What Lightning's DDPStrategy does (``_ForwardRedirection``): call the DDP wrapper -- so that its pre-forward / post-forward
    logic (bucket rebuild, reducer.prepare_for_backward) runs -- with the wrapped module's ``forward`` pointing at the step method for
    the duration of that one call."""
    orig = module.forward

    def patched(*a, **k):
        module.forward = orig
        return module.step_with_noise(*a, **k)

    module.forward = patched
    try:
        return ddp(*args, **kw)
    finally:
        module.forward = orig


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TQ_TEST_BACKEND", "gloo")
    ngpu = torch.cuda.device_count()
    dev = torch.device("cuda", rank % max(ngpu, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        from tqdne_amd.trainer import init_process_group
        init_process_group("nccl", device=dev, rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from conftest import cfg_of, load_golden, rel_err
    from tqdne_amd import LightningEDM, rng

    sd, d = load_golden("micro_unet.npz")
    cfg = dict(cfg_of(d), dropout=0.0)   # (masks are indexed by the position in the LOCAL batch: equivalence needs p = 0)
    Bg, T, steps = 8, 256, 2
    g = torch.Generator().manual_seed(11)
    sig = [0.5 * torch.randn(Bg, 3, T, generator=g) for _ in range(steps)]
    cond = [torch.randn(Bg, 5, generator=g) for _ in range(steps)]
    eps = [torch.randn(Bg, generator=g) for _ in range(steps)]
    noise = [torch.randn(Bg, 3, T, generator=g) for _ in range(steps)]
    per = Bg // world
    mine = slice(rank * per, (rank + 1) * per)

    def make():
        m = LightningEDM(cfg, {"learning_rate": 1e-3, "max_steps": 10, "eta_min": 0.0})
        m.unet.load_state_dict(sd)
        return m.to(dev).train()

    rng.seed_rank(0, rank)
    # ---- one rank, the full global batch: the reference gradients and weights
    full = make()
    opt_full = full.configure_optimizers()["optimizer"]
    g_full = []
    for s in range(steps):
        opt_full.zero_grad(set_to_none=True)
        loss_full = full.step_with_noise(sig[s].to(dev), eps[s].to(dev), noise[s].to(dev), cond=cond[s].to(dev))
        loss_full.backward()
        if s == 0:
            loss_full0 = float(loss_full)   # (compared at the first step: from the second on the weights differ by Adam's sign noise)
        g_full.append({n: p.grad.detach().clone() for n, p in full.named_parameters() if p.grad is not None})
        opt_full.step()
    w_full = {n: p.detach().clone() for n, p in full.named_parameters()}
    loss_full = loss_full0

    res = {}
    variants = {"default": {}, "bucket_view": dict(gradient_as_bucket_view=True), "static_graph": dict(static_graph=True),
                "small_buckets": dict(bucket_cap_mb=0.05)}
    for name, kw in variants.items():
        m = make()
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[dev.index], **kw)
        opt = m.configure_optimizers()["optimizer"]
        worst = flat = 0.0
        for s in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = redirected_step(ddp, m, sig[s][mine].to(dev), eps[s][mine].to(dev), noise[s][mine].contiguous().to(dev),
                                   cond=cond[s][mine].to(dev))
            loss.backward()
            torch.cuda.synchronize()
            if s == 0:
                loss0 = float(loss)
            gmax = max(float(v.abs().max()) for v in g_full[s].values())
            got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
            missing = [n for n, v in got.items() if v is None]
            assert not missing, f"{name}: parameters without a gradient after loss.backward() under DDP: {missing[:5]}"
            if s == 0:   # (the second step starts from weights that already differ by Adam's sign noise: gradients compared at the first)
                a = torch.cat([got[n].reshape(-1) for n in g_full[s]])
                b = torch.cat([g_full[s][n].reshape(-1) for n in g_full[s]])
                flat = max(flat, rel_err(a.cpu(), b.cpu()))
                for n, ref in g_full[s].items():   # per tensor against its own scale (floored: exactly-zero gradients are rounding noise)
                    worst = max(worst, float((got[n] - ref).abs().max()) / max(float(ref.abs().max()), 1e-4 * gmax))
            opt.step()
        torch.cuda.synchronize()
        chk = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).double().cpu()
        sums = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(sums, torch.stack([chk.sum(), (chk * chk).sum()]))
        losses = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(losses, torch.tensor([loss0], dtype=torch.float64))
        # (Adam's first steps move every weight by ~lr * sign(g): an element whose true gradient is ~0 -- rounding noise on both sides --
        # may step the other way, so the comparison with the one-rank run is a fraction of elements, not a maximum)
        dw = torch.cat([(p.detach() - w_full[n]).abs().reshape(-1) for n, p in m.named_parameters()])
        w_err = float((dw > 1e-5).float().mean())
        res[name] = dict(err_flat=flat, err_worst_tensor=worst, replicas_equal=bool(all(torch.equal(x, sums[0]) for x in sums)),
                         loss_mean=float(sum(l.item() for l in losses) / world), loss_full=loss_full, weights_vs_full=w_err)
        del ddp
    if rank == 0:
        print("TORCH_DDP_RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
