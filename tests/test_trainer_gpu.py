"""DataParallelTrainer on the GPU (one rank): the fused one-launch Adam must drive the SAME training trajectory as
torch.optim.Adam (edm.py:240-251).  Guards the packed-weight cache: the fused launch writes parameters through raw pointers
and has to invalidate the engines' packed MFMA fragments (engine.py repack / repack_transposed), else every step after the
first runs on stale conv weights."""

import pytest
import torch

from conftest import cfg_of, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _run(fused, steps=4):
    from tqdne_amd import LightningEDM, rng
    from tqdne_amd.trainer import DataParallelTrainer
    sd, d = load_golden("micro_unet.npz")
    cfg = dict(cfg_of(d), dropout=0.1)
    dev = torch.device("cuda:0")
    edm = LightningEDM(cfg, {"learning_rate": 2e-3, "max_steps": 50, "eta_min": 0.0}, num_sampling_steps=4)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev).train()
    rng.seed_rank(123, 0)  # same eps / noise draws and the same dropout masks in both runs
    tr = DataParallelTrainer(edm, world_size=1, fused_optimizer=fused)
    g = torch.Generator().manual_seed(5)
    batch = {"signal": (0.5 * torch.randn(4, 3, 256, generator=g)).to(dev), "cond": torch.randn(4, 5, generator=g).to(dev)}
    losses = [float(tr.train_step(batch)) for _ in range(steps)]
    edm.eval()
    with torch.no_grad():
        y = edm(batch["signal"], torch.full((4,), 0.7, device=dev), None, batch["cond"]).cpu()
    w = {k: v.detach().cpu().clone() for k, v in edm.unet.state_dict().items()}
    return losses, y, w


def test_fused_and_torch_adam_train_the_same_trajectory():
    l_f, y_f, w_f = _run(True)
    l_t, y_t, w_t = _run(False)
    print("losses fused", l_f, "torch", l_t)
    assert l_f[0] == pytest.approx(l_t[0], rel=1e-6)  # same start
    for a, b in zip(l_f, l_t):
        assert a == pytest.approx(b, rel=2e-4)
    # the learning rate is large enough that four steps move the conv weights visibly: a forward on stale packed weights
    # would differ from the torch-Adam run by far more than the tolerance
    moved = rel_err(w_f["input_blocks.1.0.in_layers.2.weight"], load_golden("micro_unet.npz")[0]["input_blocks.1.0.in_layers.2.weight"], elem=False)
    assert moved > 1e-2, moved
    assert rel_err(y_f, y_t) < 1e-3
    # Adam turns a gradient that is rounding noise (a conv bias in front of a one-channel-per-group GroupNorm) into an update
    # of +-lr whatever its size, so those few tensors may differ between two correct implementations; the rest must agree
    errs = {k: rel_err(w_f[k], w_t[k], elem=False) for k in w_f}   # (two trajectories, counted below: not a parity reference)
    close = sum(e < 2e-3 for e in errs.values())
    print("weights within 2e-3:", close, "of", len(errs), "; worst", max(errs.items(), key=lambda kv: kv[1]))
    assert close >= 0.9 * len(errs)
    assert errs["input_blocks.1.0.in_layers.2.weight"] < 2e-3 and errs["out.2.weight"] < 2e-3


def test_packed_weights_follow_the_fused_optimizer():
    """after each fused step the engine's packed conv weights equal a fresh pack of the updated parameters"""
    from tqdne_amd import LightningEDM, rng
    from tqdne_amd.trainer import DataParallelTrainer
    sd, d = load_golden("micro_unet.npz")
    cfg = cfg_of(d)
    dev = torch.device("cuda:0")
    edm = LightningEDM(cfg, {"learning_rate": 1e-2, "max_steps": 50, "eta_min": 0.0})
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev).train()
    rng.seed_rank(1, 0)
    tr = DataParallelTrainer(edm, world_size=1, fused_optimizer=True)
    g = torch.Generator().manual_seed(6)
    batch = {"signal": (0.5 * torch.randn(2, 3, 256, generator=g)).to(dev), "cond": torch.randn(2, 5, generator=g).to(dev)}
    eng = edm.unet._engine(2, 256, dev)
    for _ in range(2):
        tr.train_step(batch)
    tr.train_step(batch)  # this step's forward repacked from the parameters of step 2
    torch.cuda.synchronize()
    site = eng.conv_sites[0]
    before = site.packed.clone()
    stream = torch.cuda.current_stream(dev).cuda_stream
    eng.repack(stream)  # parameters changed in step 3 -> must repack without force
    torch.cuda.synchronize()
    assert not torch.equal(before, site.packed), "packed weights did not follow the optimizer update"


def test_plans_share_one_packed_weight_store():
    """every plan of a model (any batch, length, lane) reads the same packed fragments; a weight update is re-packed once, by the
    first plan that runs after it, and seen by all of them -- also across streams"""
    from tqdne_amd import UNetModel
    sd, d = load_golden("micro_unet.npz")
    dev = torch.device("cuda:0")
    m = UNetModel(**cfg_of(d))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    e_a, e_b, e_c = m._engine(2, 256, dev), m._engine(4, 248, dev), m._engine(2, 256, dev, lane=1)
    assert e_a.store is e_b.store is e_c.store
    for sa, sb, sc in zip(e_a.conv_sites, e_b.conv_sites, e_c.conv_sites):
        assert sa.packed.data_ptr() == sb.packed.data_ptr() == sc.packed.data_ptr(), sa.name
    g = torch.Generator().manual_seed(3)
    x, t, c = torch.randn(2, 3, 256, generator=g).to(dev), torch.rand(2, generator=g).to(dev), torch.randn(2, 5, generator=g).to(dev)
    x4 = torch.randn(4, 3, 248, generator=g).to(dev)
    t4, c4 = torch.rand(4, generator=g).to(dev), torch.randn(4, 5, generator=g).to(dev)
    with torch.no_grad():
        y0 = m(x, t, c)
        gen0 = e_a.store.gen
        m(x4, t4, c4)
        assert e_a.store.gen == gen0, "a second plan must not re-pack unchanged weights"
        for p in m.parameters():
            p.mul_(1.25)   # in-place: bumps the version counters
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            y4 = e_b.forward(x4, t4, c4, infer=True).clone()          # packs on the side stream
        y1 = e_a.forward(x, t, c, infer=True).clone()                # main stream: must wait for that pack
        torch.cuda.synchronize()
        # (one pack by the side-stream plan; the main-stream plan only adds the buffers the other plan does not have: the two-phase
        # up-sampling weights exist for lengths that are multiples of 128 only)
        assert gen0 + 1 <= e_a.store.gen <= gen0 + 2
    m2 = UNetModel(**cfg_of(d))
    m2.load_state_dict({k: v * 1.25 if v.is_floating_point() else v for k, v in sd.items()})
    m2 = m2.to(dev).eval()
    with torch.no_grad():
        r1 = m2._engine(2, 256, dev).forward(x, t, c, infer=True).clone()     # (the same inference launch lists as y1 / y4)
        r4 = m2._engine(4, 248, dev).forward(x4, t4, c4, infer=True).clone()
    assert torch.equal(y1, r1) and torch.equal(y4, r4)
    assert not torch.equal(y0, y1)


def test_range_guard_in_a_training_loop_drops_the_offending_step():
    """The deferred range guard of the fp16-range forward scheme in TRAINING (engine._range_poll + tq_adam_ema_step_guarded):
    a residual stream is blown up to the fp16 range in the middle of a run.  The step whose forward raises the flag must not be
    applied (the optimizer launch is predicated on the flag on the device -- no host sync), every later step until the host
    has moved the plans is dropped too, then training continues on bf16x3 with finite weights."""
    import warnings

    from tqdne_amd import LightningEDM, rng, tiny_1d_unet_config
    from tqdne_amd.engine import shared_range_flag
    from tqdne_amd.trainer import DataParallelTrainer
    from test_hip_unet import perturbed_state
    cfg = dict(tiny_1d_unet_config(), model_channels=64, channel_mult=(2, 2), num_res_blocks=1, dropout=0.1)  # fp16-range launches
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-3, "max_steps": 50, "eta_min": 0.0})
    edm.unet.load_state_dict(perturbed_state(edm.unet, 3))
    edm = edm.to(dev).train()
    rng.seed_rank(7, 0)
    tr = DataParallelTrainer(edm, world_size=1, fused_optimizer=True)
    g = torch.Generator().manual_seed(8)
    B, T = 2, 1024
    batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)}
    if cfg.get("cond_features"):
        batch["cond"] = torch.randn(B, 5, generator=g).to(dev)
    eng = edm.unet._engine(B, T, dev)
    assert eng.scheme == "auto" and any(d.wfmt != 0 for d, _, _ in eng._wfmt_sites), "the test net must have fp16-range launches"
    flag = shared_range_flag(edm.unet, dev)
    snap = lambda: torch.cat([p.detach().reshape(-1) for p in edm.unet.parameters()]).clone()
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and eng.scheme == "auto"
    # blow up the residual stream behind the first ResBlock (its second conv then writes values of ~1e5)
    with torch.no_grad():
        for n, p in edm.unet.named_parameters():
            if n.startswith("input_blocks.1.0.out_layers.3."):
                p.mul_(3.0e5)
    w_before = snap()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        tr.train_step(batch)                       # raises the flag; the host does not know yet
        torch.cuda.synchronize()
        assert eng.scheme == "auto", "the guard is deferred in training: the plan moves at a later step"
        assert torch.equal(snap(), w_before), "the step that raised the range flag must not be applied"
        moved_at = None
        for k in range(1, 6):                      # the host sees the flag at the start of one of the next forwards
            tr.train_step(batch)
            torch.cuda.synchronize()
            if eng.scheme == "bf16x3" and moved_at is None:
                moved_at = k
        assert moved_at is not None and moved_at <= 2, moved_at
        assert any("fp16 range" in str(w.message) for w in rec)
    w_after = snap()
    assert torch.isfinite(w_after).all(), "no inf / NaN may reach the weights"
    assert not torch.equal(w_after, w_before), "training continues once the plans are on bf16x3"
    assert all(d.wfmt == 0 for d, _, _ in eng._wfmt_sites)


def test_graph_replayed_sampler_follows_weight_updates():
    """the sampler's HIP graphs read the packed weight fragments at fixed addresses: after optimizer steps a replay must run on the
    NEW weights (re-packed before the replay), i.e. equal the eager sampler bit for bit"""
    from tqdne_amd import LightningEDM, rng
    from tqdne_amd.trainer import DataParallelTrainer
    sd, d = load_golden("micro_unet.npz")
    cfg = cfg_of(d)
    dev = torch.device("cuda:0")
    edm = LightningEDM(cfg, {"learning_rate": 5e-3, "max_steps": 50, "eta_min": 0.0}, num_sampling_steps=6)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev)
    rng.seed_rank(3, 0)
    tr = DataParallelTrainer(edm, world_size=1, fused_optimizer=True)
    g = torch.Generator().manual_seed(7)
    batch = {"signal": (0.5 * torch.randn(2, 3, 256, generator=g)).to(dev), "cond": torch.randn(2, 5, generator=g).to(dev)}
    sig = edm.edm.sampling_sigmas(6).to(dev)
    eps = (torch.randn(2, 3, 256, generator=g, dtype=torch.float64) * float(sig[0])).to(dev)
    edm.eval()
    first = {m: edm.sample_deterministically(eps, sig, None, batch["cond"], use_graph=m).clone() for m in (True, "denoiser")}   # capture
    for _ in range(3):
        edm.train()
        tr.train_step(batch)
    edm.eval()
    eager = edm.sample_deterministically(eps, sig, None, batch["cond"], use_graph=False)
    for mode in (True, "denoiser"):
        again = edm.sample_deterministically(eps, sig, None, batch["cond"], use_graph=mode)
        assert not torch.equal(again, first[mode]), "three optimizer steps at lr 5e-3 must move the sample"
        assert torch.equal(again, eager), f"graph mode {mode!r} replayed stale packed weights"
