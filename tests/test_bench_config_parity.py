"""Parity at the configuration bench.py measures (BASELINE configs[1]: paper UNet, B = 64, 3 x 4096, 4 sampler lanes on 4
streams, f16+mx6 forward convs, dropout 0.1 in the train step) -- GPU vs the CPU oracle, 1e-3 relative (max|a-b| / max|b|):

  (i)   B = 64: lanes = 4 and lanes = 1 integrate bit-identical samples; samples {0, 17, 33, 63} vs the oracle after 2 Heun steps
  (ii)  B = 2: the full 18-step / 35-NFE sample vs the oracle, error growth printed after steps 1 / 9 / 18
  (iii) the fp16-range conv scheme on trained-like statistics (heavy-tailed weights, residual-stream magnitudes up to 1e4, per-channel
        scale spread of 1e3), and the range guard that moves a plan to bf16x3 before the fp16 range is left
  (iv)  a dropout-ON training step: the kernels' counter-based masks are rebuilt on the CPU from the same hash (csrc/common.hpp
        drop_hash) and fed to the oracle; loss and every gradient must agree (edm.py:126-134, unet.py:101)
  (v)   LightningEDM.forward under autograd (edm.py:105-113 is an ordinary differentiable call in the reference)
"""

import warnings

import numpy as np
import pytest
import torch

from conftest import grad_err, rel_err
from test_hip_unet import perturbed_state

pytestmark = pytest.mark.gpu

TOL = 1e-3


def dev():
    return torch.device("cuda:0")


def _paper_edm(num_steps=18, dropout=None, seed=17):
    from tqdne_amd import LightningEDM, paper_1d_unet_config
    cfg = paper_1d_unet_config()
    if dropout is not None:
        cfg = dict(cfg, dropout=dropout)
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=num_steps)
    sd = perturbed_state(edm.unet, seed)
    edm.unet.load_state_dict(sd)
    return edm.to(dev()), sd, cfg


# ---------------------------------------------------------------------------------------------------------------- (i)
def test_bench_batch_lanes_bit_identical_and_vs_oracle():
    from oracle import edm as OE
    edm, sd, cfg = _paper_edm()
    edm.eval()
    B, T, nsteps = 64, 4096, 2
    g = torch.Generator().manual_seed(1234)
    start = torch.randn(B, 3, T, generator=g, dtype=torch.float64)
    cond = torch.randn(B, 5, generator=g)
    sig = OE.sampling_sigmas(OE.EDMParams(), 18)
    eps = (start * sig[0]).to(dev())
    s2 = sig[: nsteps + 1].to(dev())
    y4 = edm.sample_deterministically(eps, s2, None, cond.to(dev()), lanes=4)
    y1 = edm.sample_deterministically(eps, s2, None, cond.to(dev()), lanes=1)
    assert torch.equal(y4, y1), "4 lanes x 16 samples on 4 streams must integrate exactly what one lane of 64 does"
    pick = [0, 17, 33, 63]
    net = OE.make_net({"unet." + k: v for k, v in sd.items()}, cfg)
    trace = {}
    with torch.no_grad():
        OE.sample_deterministic(OE.EDMParams(), net, start[pick], 18, cond=cond[pick], trace=trace, stop_after=nsteps)
    e = rel_err(y4[pick].cpu(), trace[nsteps])
    print(f"paper UNet, B=64, 4 lanes: samples {pick} after {nsteps} Heun steps vs oracle: {e:.2e}")
    assert e < TOL


# ---------------------------------------------------------------------------------------------------------------- (ii)
def test_paper_config_full_18_step_sample_vs_oracle():
    from oracle import edm as OE
    edm, sd, cfg = _paper_edm()
    edm.eval()
    B, T = 2, 4096
    g = torch.Generator().manual_seed(99)
    start = torch.randn(B, 3, T, generator=g, dtype=torch.float64)
    cond = torch.randn(B, 5, generator=g)
    sig = OE.sampling_sigmas(OE.EDMParams(), 18)
    net = OE.make_net({"unet." + k: v for k, v in sd.items()}, cfg)
    trace = {}
    with torch.no_grad():
        ref = OE.sample_deterministic(OE.EDMParams(), net, start, 18, cond=cond, trace=trace)
    eps = (start * sig[0]).to(dev())
    errs = {}
    for n in (1, 9):
        st = edm.sample_deterministically(eps, sig[: n + 1].to(dev()), None, cond.to(dev()))
        errs[n] = rel_err(st.cpu(), trace[n])
    out = edm.sample_deterministically(eps, sig.to(dev()), None, cond.to(dev()))
    errs[18] = rel_err(out.cpu(), ref)
    print("paper UNet 3 x 4096, 18-step Heun sample (35 NFE) vs oracle: error after step 1 / 9 / 18: "
          + " / ".join(f"{errs[n]:.2e}" for n in (1, 9, 18)))
    assert all(v < TOL for v in errs.values())


# ---------------------------------------------------------------------------------------------------------------- (iii)
def _heavy(shape, g, tail_every, tail, scale=1.0):
    x = torch.randn(shape, generator=g) * scale
    m = torch.rand(shape, generator=g) < (1.0 / tail_every)
    return torch.where(m, x * tail, x)


@pytest.mark.parametrize("scheme", ["f16mx8", "f16mx6"])
@pytest.mark.parametrize("case", ["normalised", "residual_stream", "fused_skip"])
def test_f16_mx8_conv_on_trained_like_statistics(case, scheme):
    """|x| up to ~1e4 with a per-channel scale spread of 1e3, |w| up to ~10 with heavy tails: the fp16-range scheme must stay inside
    1e-3 (its fp8 corrections saturate gracefully: beyond their range the product keeps fp16's 2^-12 relative accuracy)"""
    from tqdne_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed({"normalised": 1, "residual_stream": 2, "fused_skip": 3}[case])
    B, T, Ci, Co, K = 2, 512, 256, 256, 5
    w = _heavy((Co, Ci, K), g, 40, 12.0, 0.8)                        # |w| up to ~10, heavy tails
    chan = torch.logspace(-1.5, 1.5, Ci)[torch.randperm(Ci, generator=g)]   # per-channel scale spread 1e3
    x = _heavy((B, T, Ci), g, 60, 8.0) * chan * 30.0                # |x| up to ~1e4
    assert float(x.abs().max()) < 6.0e4
    bias = torch.randn(Co, generator=g)
    kw = {}
    ref_x = x
    if case == "normalised":   # GN + SiLU prologue (what 44 of the UNet's convs see), trained-like per-channel gamma spread
        gs = (torch.randn(B, Ci, generator=g) * 0.5 + 1.0) / chan / 30.0 * torch.logspace(-1, 1, Ci)
        gh = torch.randn(B, Ci, generator=g) * 0.3
        kw = dict(gscale=gs.to(d), gshift=gh.to(d), silu=True)
        ref_x = torch.nn.functional.silu(x * gs[:, None, :] + gh[:, None, :])
    skip = None
    if case == "fused_skip":   # conv2 of a ResBlock with the raw block input riding in as the 1x1 skip conv
        Cs = 128
        xs = _heavy((B, T, Cs), g, 60, 8.0) * 300.0
        ws = _heavy((Co, Cs, 1), g, 40, 12.0, 0.8)
        bs = torch.randn(Co, generator=g)
        gs = (torch.randn(B, Ci, generator=g) * 0.5 + 1.0) / chan / 30.0
        gh = torch.randn(B, Ci, generator=g) * 0.3
        kw = dict(gscale=gs.to(d), gshift=gh.to(d), silu=True)
        ref_x = torch.nn.functional.silu(x * gs[:, None, :] + gh[:, None, :])
        skip = (xs.to(d), None, ws.to(d), bs.to(d))
    wfmt = _lib.TQ_WFMT_F16_MX8 if scheme == "f16mx8" else _lib.TQ_WFMT_F16_MX6
    y, _ = ops.conv1d(x.to(d), w.to(d), bias.to(d), wfmt=wfmt, skip=skip, **kw)
    ref = torch.nn.functional.conv1d(ref_x.double().permute(0, 2, 1), w.double(), bias.double(), padding=K // 2)
    if skip is not None:
        ref = ref + torch.nn.functional.conv1d(xs.double().permute(0, 2, 1), ws.double(), bs.double())
    # (element-wise criterion for the default scheme; round 1's f16+mx8, kept selectable, clamps its fp8 corrections on the raw
    # residual stream: 3.5e-4 norm-wise but 1.6e-3 element-wise on this stress input, which is why f16+mx6 superseded it)
    e = rel_err(y.cpu().permute(0, 2, 1), ref, elem=(scheme == "f16mx6"))
    print(f"{scheme} on trained-like statistics ({case}): max|x| {float(x.abs().max()):.3g}, max|w| {float(w.abs().max()):.3g}, "
          f"rel err {e:.2e}")
    assert torch.isfinite(y).all() and e < TOL


def test_range_guard_moves_the_plan_to_bf16x3():
    """a residual stream that approaches the fp16 range: the conv epilogues raise the guard flag (sum of squares per 128
    positions >= (65504 / 2)^2), the plan is re-packed as bf16x3 and the public forward repeats itself -- result vs oracle"""
    from oracle import unet as OU
    from tqdne_amd import UNetModel, tiny_1d_unet_config
    cfg = dict(tiny_1d_unet_config(), model_channels=64, channel_mult=(2, 2), num_res_blocks=1)  # 128-channel levels: mx8 launches
    torch.manual_seed(0)
    m = UNetModel(**cfg)
    sd = perturbed_state(m, 3)
    # blow up the residual stream behind the first ResBlock: its second conv writes values of ~1e5
    for k in list(sd):
        if k.startswith("input_blocks.1.0.out_layers.3."):
            sd[k] = sd[k] * 3.0e5
    m.load_state_dict(sd)
    m = m.to(dev()).eval()
    g = torch.Generator().manual_seed(4)
    B, T = 2, 1024
    x, t = torch.randn(B, 3, T, generator=g), torch.randn(B, generator=g) * 0.5
    eng = m._engine(B, T, dev())
    from tqdne_amd import _lib
    if _lib.requested_scheme() == "bf16x3":
        pytest.skip("TQDNE_CONV_SCHEME=bf16x3: no fp16-range launches, nothing for the guard to move")
    assert eng.scheme == "auto" and any(d.wfmt != 0 for d, _, _ in eng._wfmt_sites), "the test net must have fp16-range launches"
    with warnings.catch_warnings(record=True) as rec, torch.no_grad():
        warnings.simplefilter("always")
        y = m(x.to(dev()), t.to(dev())).cpu()
        yo = OU.unet_forward(sd, cfg, x, t, None)
    assert eng.scheme == "bf16x3" and all(d.wfmt == 0 for d, _, _ in eng._wfmt_sites)
    assert any("fp16 range" in str(w.message) for w in rec)
    e = rel_err(y, yo)
    print(f"range guard: residual stream max {float(yo.abs().max()):.3g}; after the fallback rel err vs oracle {e:.2e}")
    assert torch.isfinite(y).all() and e < TOL
    assert m._engine(B, 2 * T, dev()).scheme == "bf16x3"  # plans built later start on bf16x3


# ---------------------------------------------------------------------------------------------------------------- (iv)
def _mix32(x):
    """csrc/common.hpp mix32 on numpy uint32 arrays (wrapping arithmetic)"""
    M = np.uint32
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> M(16)
        x *= M(0x7FEB352D)
        x ^= x >> M(15)
        x *= M(0x846CA68B)
        x ^= x >> M(16)
    return x


def _drop_key(seed, site, b):
    """csrc/common.hpp drop_key: the per-sample key of dropout site ``site`` under the 64-bit ``seed``"""
    M = np.uint32
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        k = _mix32(M(seed & 0xFFFFFFFF) ^ M(0x9E3779B9))
        k = _mix32(k ^ M(seed >> 32))
        k = _mix32(k + M(0x85EBCA6B) * M(site + 1))
        return _mix32(k + M(0xC2B2AE35) * M(b + 1)), _mix32((k ^ M(0x27D4EB2F)) + M(0x165667B1) * M(b + 1))


def _drop_hash(key, e):
    """csrc/common.hpp drop_hash: the finaliser with the key's second word added between its two multiplies"""
    M = np.uint32
    ka, kb = key
    x = np.asarray(e, dtype=np.uint32) ^ ka
    with np.errstate(over="ignore"):
        x ^= x >> M(16)
        x *= M(0x7FEB352D)
        x += kb
        x ^= x >> M(15)
        x *= M(0x846CA68B)
        x ^= x >> M(16)
    return x


def _dropout_masks(cfg, B, T_of_block, seed, p):
    """{block name: (B, C, T) mask scaled by 1/(1-p)} exactly as conv1d_mfma.hip's ACT == 3 prologue draws it: element t*C + c
    of sample b at site k (k-th ResBlock in execution order, from 1) is kept iff drop_hash(drop_key(seed, k, b), t*C + c) >=
    uint32(p * 2^32)."""
    from oracle import unet as OU
    p32 = np.float32(p)
    thresh = np.uint32(int(float(p32) * 4294967296.0))
    scale = np.float32(1.0) / (np.float32(1.0) - p32)
    masks = {}
    for k, (name, C) in enumerate(OU.res_block_names(cfg), start=1):
        T = T_of_block(name)
        idx = np.arange(T * C, dtype=np.uint32)
        keep = np.stack([_drop_hash(_drop_key(seed, k, b), idx) >= thresh for b in range(B)])
        masks[name] = torch.from_numpy(np.where(keep, scale, np.float32(0)).astype(np.float32).reshape(B, T, C)).permute(0, 2, 1).contiguous()
    return masks


@pytest.mark.parametrize("which,B,T", [("tiny", 2, 4096), ("paper", 1, 1024)])
def test_dropout_on_training_step_vs_oracle(which, B, T):
    from oracle import edm as OE
    from tqdne_amd import LightningEDM, paper_1d_unet_config, rng, tiny_1d_unet_config
    p = 0.1
    cfg = dict(paper_1d_unet_config() if which == "paper" else tiny_1d_unet_config(), dropout=p)
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
    sd = perturbed_state(edm.unet, 29)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev()).train()
    g = torch.Generator().manual_seed(5)
    sig = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g) if cfg["cond_features"] else None
    eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, T, generator=g)
    rng.seed_rank(2024, 0)
    seed = rng.dropout_seed_for(torch.initial_seed(), 0, 1)   # the seed the next training forward will draw
    loss = edm.step_with_noise(sig.to(dev()), eps.to(dev()), noise.to(dev()), cond=cond.to(dev()) if cond is not None else None)
    loss.backward()
    # resolution of every ResBlock: T halves behind each Downsample of the input path and doubles behind each Upsample
    eng = edm.unet._engine(B, T, dev())
    T_by_name = {}
    res = [t for kind, t in eng.tape if kind == "res"]
    from oracle import unet as OU
    for (name, _), t in zip(OU.res_block_names(cfg), res):
        T_by_name[name] = t["out"].T
    masks = _dropout_masks(cfg, B, lambda n: T_by_name[n], seed, p)
    kept = float(np.mean([float((m > 0).float().mean()) for m in masks.values()]))
    assert abs(kept - (1 - p)) < 5e-3
    params = {("unet." + k): v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    lo = OE.loss_step(OE.EDMParams(), OE.make_net(params, cfg, dropout_masks=masks), sig, eps, noise, cond=cond)
    lo.backward()
    e_loss = rel_err(loss.detach().cpu(), lo.detach())
    gmax = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    worst, wname = 0.0, ""
    for name, q in edm.unet.named_parameters():
        if not q.requires_grad:
            continue
        ref = params["unet." + name].grad
        e = grad_err(q.grad, ref, gmax, name)
        if e > worst:
            worst, wname = e, name
    print(f"{which}: dropout {p} ON, masks rebuilt from the hash (kept {kept:.4f}): loss rel err {e_loss:.2e}; worst gradient rel err "
          f"{worst:.2e} at {wname}")
    assert e_loss < TOL and worst < TOL


# ---------------------------------------------------------------------------------------------------------------- (v)
def test_edm_forward_is_differentiable_like_the_reference():
    from oracle import edm as OE
    from tqdne_amd import LightningEDM, tiny_1d_unet_config
    cfg = dict(tiny_1d_unet_config(), dropout=0.0)
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
    sd = perturbed_state(edm.unet, 31)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev()).train()
    g = torch.Generator().manual_seed(6)
    B, T = 2, 1024
    x = torch.randn(B, 3, T, generator=g)
    sigma = torch.tensor([0.3, 7.0])
    G = torch.randn(B, 3, T, generator=g)
    y = edm(x.to(dev()), sigma.to(dev()))
    assert y.requires_grad
    (y * G.to(dev())).sum().backward()
    params = {("unet." + k): v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    yo = OE.denoise(OE.EDMParams(), OE.make_net(params, cfg), x, sigma)
    (yo * G).sum().backward()
    assert rel_err(y.detach().cpu(), yo.detach()) < TOL
    gmax = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    worst = 0.0
    for name, q in edm.unet.named_parameters():
        if q.requires_grad:
            ref = params["unet." + name].grad
            worst = max(worst, grad_err(q.grad, ref, gmax, name))
    print(f"differentiable forward: worst gradient rel err {worst:.2e}")
    assert worst < TOL


# ---------------------------------------------------------------------------------------------------------------- (vi)
def test_paper_config_consistency_sampling_b64_vs_oracle():
    """BASELINE configs[4]: consistency-model sampling on the paper UNet at the bench batch (B = 64, 3 x 4096), the 1-step
    sampler and one refinement step (consistency_model.py:63-106) -- samples {0, 17, 33, 63} vs the CPU oracle."""
    from oracle import consistency as OC
    from oracle import unet as OU
    from tqdne_amd import LithningConsistencyModel, UNetModel, paper_1d_unet_config
    cfg = paper_1d_unet_config()
    torch.manual_seed(0)
    net = UNetModel(**cfg)
    sd = perturbed_state(net, 29)
    net.load_state_dict(sd)
    cm = LithningConsistencyModel(net).to(dev()).eval()
    B, T = 64, 4096
    g = torch.Generator().manual_seed(4321)
    start = torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g)
    uni = torch.rand(B, 3, T, generator=g)
    y1 = cm.sample_from(start.to(dev()), [], [], cond=cond.to(dev()))
    y2 = cm.sample_from(start.to(dev()), [1.0], [uni.to(dev())], cond=cond.to(dev()))
    pick = [0, 17, 33, 63]
    onet = lambda x, t, c: OU.unet_forward(sd, cfg, x, t, c)
    with torch.no_grad():
        r1 = OC.sample(onet, start[pick], cond=cond[pick])
        r2 = OC.sample(onet, start[pick], [1.0], [uni[pick]], cond=cond[pick])
    e1, e2 = rel_err(y1[pick].cpu(), r1), rel_err(y2[pick].cpu(), r2)
    print(f"paper UNet, B=64, consistency sampling, samples {pick} vs oracle: 1-step {e1:.2e}, with one refinement step {e2:.2e}")
    assert e1 < TOL and e2 < TOL
    # batch independence at the bench batch: the same four waveforms sampled alone.  A plan for <= 4 samples uses the small position
    # tile for its ResBlock convs (engine.SMALL_TILE_B): same convolution arithmetic, the GroupNorm statistics summed in another
    # association order; a last-bit difference in the first GroupNorm's coefficients grows through ~50 layers to the level of the
    # path's own rounding noise against the oracle (2e-5) -> held to 1e-4; with the small tile off, bit-identical
    y1p = cm.sample_from(start[pick].to(dev()), [], [], cond=cond[pick].to(dev()))
    e_batch = rel_err(y1p.cpu(), y1[pick].cpu())
    print(f"the same four samples alone vs inside the batch of 64: {e_batch:.2e}")
    assert e_batch < 1e-4, "a sample must not depend on what else is in the batch"
    import tqdne_amd.engine as E
    old_b, old_w = E.SMALL_TILE_B, E.SMALL_TILE_WGS
    try:
        E.SMALL_TILE_B = E.SMALL_TILE_WGS = 0
        net2 = UNetModel(**cfg)
        net2.load_state_dict(sd)
        cm2 = LithningConsistencyModel(net2).to(dev()).eval()
        y1q = cm2.sample_from(start[pick].to(dev()), [], [], cond=cond[pick].to(dev()))
    finally:
        E.SMALL_TILE_B, E.SMALL_TILE_WGS = old_b, old_w
    assert torch.equal(y1q, y1[pick]), "same tiles: a sample must not depend on what else is in the batch, bit for bit"


# ---------------------------------------------------------------------------------------------------------------- (vii)
def test_paper_config_training_step_b64_all_gradients_vs_oracle():
    """The benchmarked training step at the benchmarked batch: paper UNet, B = 64, 3 x 4096, dropout off, driven the way
    DataParallelTrainer drives it (edm_loss_and_grads: one flat gradient buffer, buckets handed out from inside the sweep, the
    weight-gradient split plan of B = 64 with up to 32 units per split) -- loss and ALL parameter gradients vs torch autograd
    through the CPU oracle (edm.py:115-134).  The oracle runs the batch in chunks of 8 samples (the loss is a mean over the
    batch, so the gradient is the sum of the chunks' gradients scaled by chunk / B): bounded host memory, same result."""
    from oracle import edm as OE
    from tqdne_amd.autograd import edm_loss_and_grads
    edm, sd, cfg = _paper_edm(dropout=0.0, seed=23)
    edm.train()
    B, T, CH = 64, 4096, 8
    g = torch.Generator().manual_seed(177)
    sig = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g)
    eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, T, generator=g)
    bucket_elems = (16 << 20) // 4    # DataParallelTrainer's default: 16 MB buckets = 4 over the 62 MB of gradients
    slices = []
    loss, flat = edm_loss_and_grads(edm, sig.to(dev()), eps.to(dev()), noise.to(dev()), cond.to(dev()), None,
                                    on_bucket=lambda sl: slices.append((sl.data_ptr(), sl.numel())), bucket_elems=bucket_elems)
    torch.cuda.synchronize()
    eng = edm.unet._engine(B, T, dev())
    # the buckets tile [0, n_grad) of the flat buffer in order
    assert len(slices) == 4, [n for _, n in slices]
    at = flat.data_ptr()
    for ptr, n in slices:
        assert ptr == at
        at += 4 * n
    assert at == flat.data_ptr() + 4 * eng._bwd.n_grad
    params = {("unet." + k): v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
    net = OE.make_net(params, cfg)
    lo = 0.0
    for i in range(0, B, CH):
        sl = slice(i, i + CH)
        l = OE.loss_step(OE.EDMParams(), net, sig[sl], eps[sl], noise[sl], cond=cond[sl]) * (CH / B)
        l.backward()
        lo += float(l.detach())
    e_loss = abs(float(loss) - lo) / abs(lo)
    gmax = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    worst, wname, n = 0.0, "", 0
    for name, q in edm.unet.named_parameters():
        if not q.requires_grad:
            continue
        e = grad_err(q.grad, params["unet." + name].grad, gmax, name)
        n += 1
        if e > worst:
            worst, wname = e, name
    print(f"paper UNet, B=64 training step (flat buffer, {len(slices)} buckets): loss rel err {e_loss:.2e}; worst of {n} gradients "
          f"{worst:.2e} at {wname}")
    assert n == 310 and e_loss < TOL and worst < TOL
