"""The reference's DEFAULT per-device batch: ``-b 256`` (experiments/train_1d_edm.py:83-85) on the paper UNet, 3 x 4096 -- four times
the bench's batch.  Plans keep every activation in static buffers (``use_checkpoint=True`` shares the block-internal ones and
recomputes them in the backward, DESIGN.md section 1: measured below), so B = 256 is a memory and an index-range question:

  * one training step (dropout off so that chunks are comparable): finite loss and gradients, and the same loss / gradients as the
    mean over four B = 64 steps on the four quarters of the batch (a (b, t)-split weight gradient and a GroupNorm per sample: only
    the association of the sums differs);
  * two Heun steps of the sampler (4 lanes x 64 samples): samples {0, 100, 255} bit-identical to the same samples integrated in a
    B = 64 call (same plan shape, same tiles);
  * the peak device memory of both is printed (and bounded: the box has 288 GB);
  * the bounded plan cache (tqdne_amd/_cache.py): a 7th shape evicts the least recently used one, a re-built plan reproduces its result.
"""

import pytest
import torch

from conftest import rel_err
from test_hip_unet import dev, perturbed_state

pytestmark = pytest.mark.gpu


def _paper_edm(dropout=0.0, steps=18):
    from tqdne_amd import LightningEDM, paper_1d_unet_config
    cfg = dict(paper_1d_unet_config(), dropout=dropout)
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=steps)
    edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
    return edm.to(dev())


@pytest.mark.timeout(1200)
def test_reference_default_batch_256_train_step_and_sampler():
    from oracle import edm as OE
    edm = _paper_edm()
    B, T = 256, 4096
    g = torch.Generator().manual_seed(256)
    sig = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev())
    cond = torch.randn(B, 5, generator=g).to(dev())
    eps, noise = torch.randn(B, generator=g).to(dev()), torch.randn(B, 3, T, generator=g).to(dev())
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()

    # ---- training step at B = 256
    edm.train()
    loss = edm.step_with_noise(sig, eps, noise, cond=cond)
    loss.backward()
    torch.cuda.synchronize()
    peak_train = torch.cuda.max_memory_allocated()
    names = [n for n, p in edm.unet.named_parameters() if p.requires_grad]
    g256 = {n: p.grad.detach().clone() for n, p in edm.unet.named_parameters() if p.requires_grad}
    assert torch.isfinite(loss) and all(torch.isfinite(v).all() for v in g256.values())
    loss256 = float(loss)
    # the same step as four quarters (B = 64 plans); the B = 256 plan goes first: the quarters need its memory on smaller boxes
    for p in edm.unet.parameters():
        p.grad = None
    acc = {n: torch.zeros_like(v) for n, v in g256.items()}
    lsum = 0.0
    for q in range(4):
        s = slice(64 * q, 64 * (q + 1))
        l = edm.step_with_noise(sig[s].contiguous(), eps[s].contiguous(), noise[s].contiguous(), cond=cond[s].contiguous())
        l.backward()
        lsum += float(l)
        for n, p in edm.unet.named_parameters():
            if p.requires_grad:
                acc[n] += p.grad
                p.grad = None
    assert abs(loss256 - lsum / 4) < 1e-5 * abs(loss256), (loss256, lsum / 4)
    flat_a = torch.cat([g256[n].reshape(-1) for n in names])
    flat_b = torch.cat([acc[n].reshape(-1) / 4 for n in names])
    e_flat = rel_err(flat_a.cpu(), flat_b.cpu())
    gmax = float(flat_b.abs().max())
    worst = max(float((g256[n] - acc[n] / 4).abs().max()) / max(float(acc[n].abs().max()) / 4, 1e-4 * gmax) for n in names)
    print(f"B=256 train step: loss {loss256:.6f} (mean of quarters {lsum / 4:.6f}); gradients vs mean of four B=64 steps: flat {e_flat:.2e}, "
          f"worst tensor (own scale) {worst:.2e}; peak memory {(peak_train - base) / 2**30:.1f} GiB above the {base / 2**30:.2f} GiB of weights")
    assert e_flat < 1e-4 and worst < 1e-3

    # ---- sampler at B = 256: 4 lanes x 64
    edm.eval()
    start = torch.randn(B, 3, T, generator=g, dtype=torch.float64)
    sg = OE.sampling_sigmas(OE.EDMParams(), 18)
    s2 = sg[:3].to(dev())   # two Heun steps = 4 network evaluations
    x0 = (start * sg[0]).to(dev())
    torch.cuda.reset_peak_memory_stats()
    y = edm.sample_deterministically(x0, s2, None, cond)
    torch.cuda.synchronize()
    peak_smp = torch.cuda.max_memory_allocated()
    assert torch.isfinite(y).all()
    pick = [0, 100, 255]
    rows = [0, 36, 63]                                   # where the three samples sit in the B = 64 call
    xs, cs = x0[:64].clone(), cond[:64].clone()
    for r, p_ in zip(rows, pick):
        xs[r], cs[r] = x0[p_], cond[p_]
    y64 = edm.sample_deterministically(xs, s2, None, cs, lanes=1)
    torch.cuda.synchronize()
    for r, p_ in zip(rows, pick):
        assert torch.equal(y[p_], y64[r]), f"sample {p_} of the B = 256 call differs from the same sample in a B = 64 call"
    # and against the oracle (sample 100 only: a CPU forward of the paper UNet takes seconds per evaluation)
    sd = {k: v.detach().cpu() for k, v in edm.unet.state_dict().items()}
    net = OE.make_net({"unet." + k: v for k, v in sd.items()}, edm.config)
    trace = {}
    with torch.no_grad():
        OE.sample_deterministic(OE.EDMParams(), net, start[100:101], 18, cond=cond[100:101].cpu(), trace=trace, stop_after=2)
    e = rel_err(y[100:101].cpu(), trace[2])
    print(f"B=256 sampler (4 lanes x 64): samples {pick} bit-identical to a B=64 call; sample 100 vs oracle after 2 Heun steps {e:.2e}; "
          f"peak memory {peak_smp / 2**30:.1f} GiB")
    assert e < 1e-3
    assert peak_train < 200 * 2**30 and peak_smp < 200 * 2**30


@pytest.mark.timeout(600)
def test_batch_256_train_step_with_use_checkpoint_reports_memory():
    """the same B = 256 training step with ``use_checkpoint=True`` (block-internal activations shared and recomputed): finite, same loss
    as without, and the peak memory of the two printed side by side"""
    import gc
    from tqdne_amd import LightningEDM, paper_1d_unet_config, rng
    B, T = 256, 4096
    g = torch.Generator().manual_seed(256)
    sig = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev())
    cond = torch.randn(B, 5, generator=g).to(dev())
    eps, noise = torch.randn(B, generator=g).to(dev()), torch.randn(B, 3, T, generator=g).to(dev())
    out = {}
    for ck in (False, True):
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        edm = LightningEDM(dict(paper_1d_unet_config(), dropout=0.1, use_checkpoint=ck), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
        edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
        edm = edm.to(dev()).train()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        rng.seed_rank(5, 0)
        loss = edm.step_with_noise(sig, eps, noise, cond=cond)
        loss.backward()
        torch.cuda.synchronize()
        gsum = float(sum(p.grad.double().abs().sum() for p in edm.unet.parameters() if p.grad is not None))
        out[ck] = (float(loss.detach()), gsum, (torch.cuda.max_memory_allocated() - base) / 2**30)
        assert torch.isfinite(loss.detach()) and gsum == gsum
        del edm, loss
    print(f"B=256 train step, peak memory above weights + inputs: {out[False][2]:.1f} GiB; with use_checkpoint {out[True][2]:.1f} GiB")
    # (the loss is a float-atomics sum over 3.1 M elements: the order of the adds, hence its last bits, varies from launch to launch)
    assert abs(out[True][0] - out[False][0]) < 5e-6 * abs(out[False][0])
    assert abs(out[True][1] - out[False][1]) < 1e-4 * out[False][1]
    assert out[True][2] < out[False][2]


def test_plan_cache_evicts_and_rebuilt_plan_reproduces():
    """micro UNet, eight batch sizes through a cache of six shapes: the first two are evicted, their re-built plans give the same bits"""
    from conftest import cfg_of, load_golden
    from tqdne_amd import UNetModel, _cache
    sd, d = load_golden("micro_unet.npz")
    net = UNetModel(**cfg_of(d))
    net.load_state_dict(sd)
    net = net.to(dev()).eval()
    g = torch.Generator().manual_seed(3)
    T = 256
    outs = {}
    with torch.no_grad():
        for B in range(1, _cache.PLAN_SHAPES + 3):
            x, t, c = torch.randn(B, 3, T, generator=g).to(dev()), torch.rand(B, generator=g).to(dev()), torch.randn(B, 5, generator=g).to(dev())
            outs[B] = (x, t, c, net(x, t, c).clone())
        assert len(net._engine_cache.groups()) == _cache.PLAN_SHAPES and net._engine_cache.evictions == 2
        assert net._engine_cache.get((1, T, str(dev()), 0)) is None
        uid_before = max(e.uid for e in net._engine_cache.values())
        x, t, c, y = outs[1]
        y2 = net(x, t, c)
        assert torch.equal(y, y2)
        assert net._engine_cache.get((1, T, str(dev()), 0)).uid > uid_before   # a NEW plan (captured graphs are keyed by the uid)


def test_evicted_training_plan_gives_its_memory_back_without_a_gc_pass():
    """round-5 advisor finding (medium): a plan that has run a backward sits in a reference cycle with its backward plan, so after an
    eviction its buffers used to survive until a cyclic-GC pass -- variable-shape training leaked a plan per shape.  Here: training
    steps on PLAN_SHAPES + 2 batch sizes with the cyclic collector OFF; the allocated memory after the evictions must be what
    PLAN_SHAPES plans need, not what PLAN_SHAPES + 2 need, and the first plan must be gone."""
    import gc
    import weakref
    from conftest import cfg_of, load_golden
    from tqdne_amd import LightningEDM, _cache
    sd, d = load_golden("micro_unet.npz")
    edm = LightningEDM(cfg_of(d), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}).to(dev())
    edm.unet.load_state_dict(sd)
    edm.train()
    T = 2048
    g = torch.Generator().manual_seed(5)

    def step(B):
        batch = {"signal": torch.randn(B, 3, T, generator=g).to(dev()), "cond": torch.randn(B, 5, generator=g).to(dev())}
        loss, _ = edm.step_and_backward(batch)
        return float(loss)

    gc.collect()
    gc.disable()
    try:
        sizes = [8 * (i + 1) for i in range(_cache.PLAN_SHAPES + 2)]   # growing: the evicted plans are the SMALL ones, so the bound is strict
        step(sizes[0])
        first = weakref.ref(edm.unet._engine(sizes[0], T, dev()))
        assert first()._bwd is not None and first()._bwd.e is first()
        mem = []
        for B in sizes[1:]:
            step(B)
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
        assert edm.unet._engine_cache.evictions == 2
        assert first() is None, "the evicted training plan is still alive (plan <-> backward-plan cycle not broken)"
        # plans scale with the batch: adding B = 24 next to two live plans costs plan(24); adding B = 64 while B = 16 is evicted must
        # cost about plan(64) - plan(16), clearly less than plan(64) - plan(8)
        per_b = (mem[1] - mem[0]) / sizes[2]
        grow_last = mem[-1] - mem[-2]
        assert grow_last < per_b * (sizes[-1] - sizes[0]), (grow_last, per_b)
    finally:
        gc.enable()
