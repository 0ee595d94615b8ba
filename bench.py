#!/usr/bin/env python3
"""bench.py -- waveforms/s of the tqdne 1-D EDM hot path on MI355X (BASELINE.json metric).

One "step" = one EDM training step (zero_grad, loss, backward, gradient all-reduce for N>1, Adam + cosine LR) on a
batch of B synthetic 3 x 4096 waveforms PLUS one 18-step deterministic Heun sample (35 UNet evaluations) of a batch
of B waveforms, paper 1-D UNet (BASELINE.json configs[1]); value = N * B / t_step.  Inputs are resident in HBM
before the timed region.  One process per GPU; for N > 1 launch with torch.distributed.run (RCCL all-reduce).

Also printed in the same JSON line:
  roofline      dominant kernel (fused k=5 conv on bf16x3 MFMA): algorithmic FLOP per launch / HIP-event time per launch
  cpu_baseline  the CPU oracle (a port: our restatement, pinned to the reference by golden vectors) timed on this host
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the device is touched: see tqdne_amd/__init__.py (4 lanes + RCCL's stream)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def perturbed_state(model, seed):
    """SURVEY 8c/8d recipe: re-draw zero-initialised tensors, jitter GroupNorm affines (cost-neutral, pins parity)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        v = v.clone()
        is_gn = v.ndim == 1 and (".in_layers.0." in k or ".out_layers.0." in k or ".norm." in k or ".out.0." in k or k.startswith("out.0."))
        if is_gn and k.endswith("weight"):
            v = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif is_gn and k.endswith("bias"):
            v = 0.1 * torch.randn(v.shape, generator=g)
        elif torch.count_nonzero(v) == 0:
            v = 0.02 * torch.randn(v.shape, generator=g)
        sd[k] = v
    return sd


def conv_flops_per_sample(unet, T):
    """2*Cin*Cout*K*T_out summed over every Conv1d of the UNet + 4*H*D*T^2 per attention."""
    from tqdne_amd import engine  # noqa: F401
    total = 0
    # walk the plan of a dry engine description without touching the GPU: recompute from module shapes
    T_l = T
    def conv(c):
        return 2 * c.in_channels * c.out_channels * c.kernel_size[0]
    seq = []
    for blk in list(unet.input_blocks):
        for layer in blk:
            kind = getattr(layer, "kind", None)
            if kind is None:
                total += conv(layer) * T_l
            elif kind == "res":
                total += (conv(layer.in_layers[2]) + conv(layer.out_layers[3])) * T_l
                if not isinstance(layer.skip_connection, torch.nn.Identity):
                    total += conv(layer.skip_connection) * T_l
            elif kind == "attn":
                total += (conv(layer.qkv) + conv(layer.proj_out)) * T_l + 4 * layer.channels * T_l * T_l
            elif kind == "down":
                T_l = (T_l + 2 - 3) // 2 + 1
                total += conv(layer.op) * T_l
    for layer in unet.middle_block:
        kind = layer.kind
        if kind == "res":
            total += (conv(layer.in_layers[2]) + conv(layer.out_layers[3])) * T_l
        else:
            total += (conv(layer.qkv) + conv(layer.proj_out)) * T_l + 4 * layer.channels * T_l * T_l
    for blk in unet.output_blocks:
        for layer in blk:
            kind = layer.kind
            if kind == "res":
                total += (conv(layer.in_layers[2]) + conv(layer.out_layers[3]) + conv(layer.skip_connection)) * T_l
            elif kind == "attn":
                total += (conv(layer.qkv) + conv(layer.proj_out)) * T_l + 4 * layer.channels * T_l * T_l
            elif kind == "up":
                T_l *= 2
                total += conv(layer.conv) * T_l
    total += conv(unet.out[2]) * T_l
    return total


# SURVEY.md 8d byte model: fused-minimum activation bytes per sample and forward (every conv / attention core reads its input and
# writes its output once in fp32; everything else fused; probed with hooks over the imported reference).
# Weights W = 4 bytes x parameters, read once per call; train = 3 B A + 3 W + 7 W (Adam) + 3 W (EMA: read ema, p; write ema).
ALGO_A = {"paper": 174.7e6, "tiny": 74.3e6, "latent_unet": 175.2e6, "ae_enc": 130.7e6, "ae_dec": 176.6e6}
AE_BASE = dict(model_channels=64, channel_mult=(1, 2, 4), attention_resolutions=(), num_res_blocks=2, dims=1, conv_kernel_size=5, dropout=0.1)
EMA_DECAY = 0.999   # the reference's 1-D trainer always runs the EMA callback (experiments/train_1d_edm.py:55, tqdne/ema.py:24-28)


def step_bytes(A, W, B, nfe, train=True, sample=True, ema=True):
    """algorithmic HBM bytes of [one train step] + [one nfe-evaluation sample] at batch B"""
    return ((3 * B * A + (13 if ema else 10) * W) if train else 0) + ((nfe * (B * A + W)) if sample else 0)


def pmc_traffic_live(timeout_s=60):
    """HBM bytes of ONE launch of the dominant conv (512 -> 256, k = 5, T = 1024, B = 64) measured in this run: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section), each around tools/bench_one.py in a
    child process (started as a child, never exec'ed from this GPU-initialised process; the program itself follows ``--``).  gfx950
    corrections: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (x 2), both counters are in KiB.  None when the
    profiler is missing, this process is itself being profiled, or a pass fails -- the line then keeps the recorded file's value."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None   # (bench.py under rocprofv3, e.g. tools/profile.sh: no profiler inside a profiled process)
    out = tempfile.mkdtemp(prefix="tqdne_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_one.py"), "256", "256", "256", "5", "1024", "64", "5"]
            subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=timeout_s, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "conv1d_mfma" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        got.append(float(r["Counter_Value"]))
            if not got:
                return None
            vals[ctr] = sum(got) / len(got)
        rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
        return {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "hbm_read_bytes_corrected": rd,
                "hbm_write_bytes": wr, "hbm_traffic_bytes_per_launch": rd + wr}
    except Exception as e:   # (a reported extra: never takes the headline line down)
        log(f"live PMC pass skipped: {e!r}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def hbm_block(nbytes, ms, note=None):
    gbps = nbytes / (ms * 1e-3) / 1e9
    d = dict(bound="hbm", achieved=gbps, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbps / HBM_PEAK_GBS, algorithmic_bytes_per_step=nbytes)
    if note:
        d["note"] = note
    return d


PARITY_TRAJECTORY = (1, 6, 12)   # sampler steps whose state is compared on the way to the final sample (error growth over the NFEs)


def parity_inputs(cfg, B, T):
    """Inputs of the same-run parity gate: drawn from a seeded CPU generator, so the GPU
    process and the CPU-oracle child build bit-identical tensors without exchanging them."""
    g = torch.Generator().manual_seed(4242)
    d = dict(signal=0.5 * torch.randn(B, 3, T, generator=g),
             cond=torch.randn(B, 5, generator=g) if cfg.get("cond_features") else None,
             sigma=torch.tensor([0.02, 0.5, 5.0, 60.0] * ((B + 3) // 4))[:B],
             eps=torch.randn(B, generator=g), noise=torch.randn(B, 3, T, generator=g),
             start=torch.randn(B, 3, T, generator=g, dtype=torch.float64))
    d["noisy"] = d["signal"] + d["sigma"][:, None, None] * d["noise"]
    return d


def _cpu_baseline_worker(cfg_name, B, T, nsample_steps, seed, nthreads, reps, parity_path=None):
    """Runs in a child process: time the CPU oracle (oracle/ = our PyTorch-CPU restatement, a "port") on a bounded
    sample of the same workload: full train steps (fwd + bwd + Adam + EMA) and 18-step samples at batch B; one untimed warm-up of
    each half (a train step and a 2-step sample), then ``reps`` timed repetitions, printed one JSON line each.
    ``cfg_name``: "paper" / "tiny" (EDM on 3 x T), "latent" (BASELINE configs[3]: frozen autoencoder 3 x T <-> 16 x T / 4 + paper-shape
    latent UNet: train = encode + latent step, sample = latent sample + decode), "consistency" (configs[4]: train = nothing, sample =
    ONE network evaluation under the consistency forward)."""
    import torch
    from oracle import edm as OE
    from tqdne_amd import UNetModel, paper_1d_unet_config, tiny_1d_unet_config

    torch.set_num_threads(nthreads)
    latent = cfg_name == "latent"
    cfg = tiny_1d_unet_config() if cfg_name == "tiny" else paper_1d_unet_config(**(dict(in_channels=16, out_channels=16) if latent else {}))
    torch.manual_seed(0)
    sd = perturbed_state(UNetModel(**cfg), 17)
    g = torch.Generator().manual_seed(seed)
    sig = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g) if cfg.get("cond_features") else None
    params = {("unet." + k): v.clone().requires_grad_(v.is_floating_point() and k != "time_embed.W") for k, v in sd.items()}
    trainable = [p for p in params.values() if p.requires_grad]
    opt = torch.optim.Adam(trainable, lr=1e-4)
    ema = [q.detach().clone() for q in trainable]
    p = OE.EDMParams()
    net = OE.make_net(params, cfg)
    C_l, T_l = 3, T
    if latent:
        from oracle import autoencoder as OA
        from tqdne_amd import LightningAutoencoder
        enc_cfg, dec_cfg = dict(AE_BASE, in_channels=3, out_channels=32), dict(AE_BASE, in_channels=16, out_channels=3)
        ae_sd = perturbed_state(LightningAutoencoder(enc_cfg, dec_cfg, {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0}), 19)
        C_l, T_l = 16, T // 4

    def train():
        if cfg_name == "consistency":
            return
        x = sig
        if latent:
            with torch.no_grad():
                x = OA.encode(ae_sd, enc_cfg, sig, torch.randn(B, C_l, T_l, generator=g))[0]
        opt.zero_grad()
        loss = OE.loss_step(p, net, x, torch.randn(B, generator=g), torch.randn(B, C_l, T_l, generator=g), cond=cond)
        loss.backward()
        opt.step()
        with torch.no_grad():   # tqdne/ema.py:24-28
            torch._foreach_lerp_(ema, trainable, 1 - EMA_DECAY)

    def sample(n):
        with torch.no_grad():
            if cfg_name == "consistency":
                from oracle import consistency as OC
                OC.sample(net, torch.randn(B, 3, T, generator=g), cond=cond)
                return
            z = OE.sample_deterministic(p, net, torch.randn(B, C_l, T_l, generator=g, dtype=torch.float64), n, cond=cond)
            if latent:
                OA.decode(ae_sd, dec_cfg, z.float())

    if parity_path:
        # same-run parity gate: the oracle's denoiser output, loss (dropout off) and full sample for the injected inputs, on the
        # INITIAL weights (before the first optimizer step below); doubles as the warm-up of the sampling half
        import numpy as np
        pin = parity_inputs(cfg, B, T)
        with torch.no_grad():
            den = OE.denoise(p, net, pin["noisy"], pin["sigma"], cond=pin["cond"])
            lss = OE.loss_step(p, net, pin["signal"], pin["eps"], pin["noise"], cond=pin["cond"])
            trace = {}
            smp = OE.sample_deterministic(p, net, pin["start"], nsample_steps, cond=pin["cond"], trace=trace)
        traj = {f"sample_step{k}": trace[k].numpy() for k in PARITY_TRAJECTORY if k < nsample_steps}
        np.savez(parity_path, denoise=den.numpy(), loss=lss.numpy(), sample=smp.numpy(), **traj)
        train()
    else:
        train()
        sample(2)
    print(json.dumps(dict(stage="warm")), flush=True)
    for r in range(reps):
        t0 = time.perf_counter()
        train()
        t_train = time.perf_counter() - t0
        t0 = time.perf_counter()
        sample(nsample_steps)
        t_sample = time.perf_counter() - t0
        print(json.dumps(dict(stage="rep", rep=r, t_train=t_train, t_sample=t_sample)), flush=True)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _cpu_run(cfg_name, B, T, nsample_steps, seed, nthreads, reps, timeout_s, parity_path=None):
    import subprocess
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"bench._cpu_baseline_worker({cfg_name!r}, {B}, {T}, {nsample_steps}, {seed}, {nthreads}, {reps}, {parity_path!r})")
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(nthreads))
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=timeout_s, env=env, cwd=ROOT)
        out = r.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    reps_done = [json.loads(l) for l in out.splitlines() if l.startswith("{") and '"rep"' in l]
    return reps_done


def _median(v):
    v = sorted(v)
    n = len(v)
    return None if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


def cpu_baseline(cfg_name, B, T, nsample_steps, seed, timeout_s=150, parity_path=None):
    """The CPU oracle on the host cores of this box: all cores (<= 64 threads), warm-up + median of 3
    repetitions of the same workload at batch B, and a 1-thread line at batch 1 (one repetition after the warm-up; a
    1-thread step of the paper UNet takes ~10 s per waveform).  Each leg runs in a subprocess with a hard timeout."""
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    nthreads = max(1, min(ncores, 64))
    reps = _cpu_run(cfg_name, B, T, nsample_steps, seed, nthreads, 3, timeout_s + (60 if parity_path else 0), parity_path)
    t_train, t_sample = _median([r["t_train"] for r in reps]), _median([r["t_sample"] for r in reps])
    res = dict(value=None, unit="waveforms/s", cores=nthreads, kind="port", train_s=t_train, sample_s=t_sample, reps=len(reps),
               sample=f"{cfg_name} UNet, B={B}, 3x{T}: warm-up, then median of {len(reps)} x (1 train step + 1 x {nsample_steps}-step "
                      f"sample), torch {torch.__version__} CPU, {nthreads} threads ({ncores} cores visible, {_cpu_model()})")
    if t_train is not None:
        res["value"] = B / (t_train + t_sample)
    else:
        res["sample"] += f" -- did not finish within {timeout_s} s"
    one = _cpu_run(cfg_name, 1, T, nsample_steps, seed, 1, 1, timeout_s)
    if one:
        res["one_thread"] = dict(value=1.0 / (one[0]["t_train"] + one[0]["t_sample"]), unit="waveforms/s", cores=1,
                                 train_s=one[0]["t_train"], sample_s=one[0]["t_sample"],
                                 sample=f"same workload at B=1, 1 thread, 1 repetition after a warm-up")
    else:
        res["one_thread"] = dict(value=None, sample=f"B=1, 1 thread: did not finish within {timeout_s} s")
    return res


def parity_block(gpu, cpu_path, args):
    """The same-run parity gate of the bench line: HIP path vs the CPU oracle (child process of this run) on identical injected
    inputs and the initial weights; both of SURVEY 8c(4)'s metrics per checkpoint, bar 1e-3."""
    import numpy as np
    blk = dict(reference="oracle/ (CPU restatement pinned to the reference by tests/golden), run in this bench's CPU child",
               inputs=f"{args.config} UNet, initial weights, B={args.cpu_batch}, 3x{args.length}, seeded (bench.parity_inputs)",
               tolerance=1e-3, checkpoints={}, **{"pass": None})
    if not (cpu_path and os.path.exists(cpu_path)):
        blk["error"] = "the CPU oracle child did not deliver its outputs (timeout?)"
        return blk
    z = np.load(cpu_path)
    ok = True
    names = {"denoise": "LightningEDM.forward, sigma in {0.02, 0.5, 5, 60}", "loss": "EDM loss (dropout off)",
             "sample": f"{args.sample_steps}-step Heun sample ({2 * args.sample_steps - 1} NFE)"}
    traj = [f"sample_step{k}" for k in PARITY_TRAJECTORY if k < args.sample_steps and f"sample_step{k}" in gpu]
    for k in traj:   # (error growth across the NFEs: the sampler's fp64 state after k of the steps, 2k network evaluations)
        names[k] = f"sampler state after {k[len('sample_step'):]} of the {args.sample_steps} steps ({2 * int(k[len('sample_step'):])} NFE)"
    for k in ["denoise", "loss"] + traj + ["sample"]:
        a, b = gpu[k].double().reshape(-1), torch.from_numpy(z[k]).double().reshape(-1)
        d = (a - b).abs()
        norm = float(d.max() / b.abs().max().clamp_min(1e-30))
        rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
        elem = float((d / (b.abs() + rms)).max())
        blk["checkpoints"][k] = dict(what=names[k], max_rel=norm, allclose_rtol_atol_rms=elem)
        ok = ok and norm < 1e-3 and elem <= 1e-3
    blk["pass"] = bool(ok)
    return blk


def _cpu_extra(cfg_name, B, T, args, reps, timeout_s, what):
    """bounded CPU-oracle baseline of one of the extra configurations (all visible cores, <= 64 threads): median of ``reps``
    repetitions after a warm-up; None when it does not finish"""
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    nth = max(1, min(ncores, 64))
    got = _cpu_run(cfg_name, B, T, args.sample_steps, 99, nth, reps, timeout_s)
    if not got:
        return dict(value=None, unit="waveforms/s", cores=nth, kind="port", sample=f"{what}: did not finish within {timeout_s} s")
    tt, tsm = _median([x["t_train"] for x in got]), _median([x["t_sample"] for x in got])
    return dict(value=B / (tt + tsm), unit="waveforms/s", cores=nth, kind="port", train_s=tt, sample_s=tsm,
                sample=f"{what}, B={B}: warm-up, then median of {len(got)} repetition(s), CPU oracle, {nth} threads")


def other_configs(dev, args):
    """BASELINE.json configs[0], [3], [4], the reference's real data shape and the reference's default batch next to the headline
    (configs[1]): parity-test cases, reported with the same step definition (1 train step incl. EMA + one 18-step sample; cfg4: one
    consistency sample) so that the small / latent / consistency paths have a number in the driver's record.  Every entry carries
    ``hbm_roofline`` (SURVEY.md 8d byte model over the measured time, against 8 TB/s) and, where a bounded CPU run fits, ``cpu_baseline``.
      cfg0  tiny UNet (32 base channels, no attention, unconditioned), B = 4, 3 x 4096 -- CPU oracle at B = 4 in full
      cfg3  latent EDM: frozen autoencoder 3 x 16384 <-> 16 x 4096 + paper-shape latent UNet, B = 16 and B = 64 (sample = 18 steps + decode)
      cfg4  consistency 1-step sampling, B = 64
      6 x 4064  the reference's data shape through the GPU MovingAverageEnvelope
      B = 256   the reference's default per-device batch (train_1d_edm.py:83-85), with and without use_checkpoint"""
    from tqdne_amd import LightningAutoencoder, LightningEDM, paper_1d_unet_config, rng, tiny_1d_unet_config
    from tqdne_amd.trainer import DataParallelTrainer
    res = []
    nfe = 2 * args.sample_steps - 1
    opt = {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0}
    cpu = not args.no_cpu_baseline
    nparam = lambda mod: sum(p.numel() for p in mod.parameters())

    def med(fn, n=5):
        fn(); fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append(1e3 * (time.perf_counter() - t0))
        return _median(ts)

    def release():
        """the previous configuration's model, plans and scratch are gone before the next one is built (and before a peak-memory reading
        takes its base line): modules and plans reference each other, so collect, then hand the blocks back"""
        import gc
        gc.collect()
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()

    def entry(config, B, unit, ms, ms_train, nbytes, **parts):
        return dict(config=config, value=B / (ms * 1e-3), unit=unit, ms_per_step=ms,
                    train_wf_s=B / (ms_train * 1e-3), sample_wf_s=B / ((ms - ms_train) * 1e-3),
                    parts=dict(train_ms=ms_train, sample_ms=ms - ms_train, **parts), hbm_roofline=hbm_block(nbytes, ms))

    # ---- cfg0
    try:
        torch.manual_seed(args.seed)
        cfg = tiny_1d_unet_config()
        edm = LightningEDM(cfg, opt, num_sampling_steps=args.sample_steps)
        edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
        edm = edm.to(dev)
        B, T = 4, 4096
        g = torch.Generator().manual_seed(4321)
        batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)}
        tr = DataParallelTrainer(edm, world_size=1, ema_decay=EMA_DECAY)
        sig = edm.edm.sampling_sigmas(args.sample_steps).to(dev)
        eps0 = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev) * sig[0]

        def step():
            edm.train()
            tr.train_step(batch)
            edm.eval()
            edm.sample_deterministically(eps0, sig, None, None)

        def train_only():
            edm.train()
            tr.train_step(batch)

        ms, ms_train = med(step), med(train_only)
        r = entry("cfg0: tiny 1-D EDM UNet (32 base channels, 2 res blocks, no attention), B=4, 3x4096: 1 train step + "
                  f"{args.sample_steps}-step Heun sample", B, "waveforms/s", ms, ms_train,
                  step_bytes(ALGO_A["tiny"], 4 * nparam(edm.unet), B, nfe))
        if cpu:
            r["cpu_baseline"] = _cpu_extra("tiny", B, T, args, 2, 120, f"tiny UNet, 3x{T} in full (1 train step + 1 x {args.sample_steps}-step sample)")
        res.append(r)
        del edm, tr
    except Exception as e:   # (a reported extra: never takes the headline line down)
        res.append(dict(config="cfg0", error=repr(e)))
    # ---- cfg3
    try:
        release()
        torch.manual_seed(args.seed)
        ae = LightningAutoencoder(dict(AE_BASE, in_channels=3, out_channels=32), dict(AE_BASE, in_channels=16, out_channels=3),
                                  {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0})
        ae.load_state_dict(perturbed_state(ae, 19))
        ae = ae.to(dev).eval()
        edm = LightningEDM(paper_1d_unet_config(in_channels=16, out_channels=16), opt, num_sampling_steps=args.sample_steps, autoencoder=ae)
        edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
        edm = edm.to(dev)
        T = 16384
        tr = DataParallelTrainer(edm, world_size=1, ema_decay=EMA_DECAY)
        W_u, W_e, W_d = 4 * nparam(edm.unet), 4 * nparam(ae.encoder), 4 * nparam(ae.decoder)
        for B in (16, 64):
            g = torch.Generator().manual_seed(4322)
            batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev), "cond": torch.randn(B, 5, generator=g).to(dev)}

            def step3():
                edm.train()
                tr.train_step(batch)
                edm.eval()
                edm.sample((B, 3, T), cond=batch["cond"])

            def train3():
                edm.train()
                tr.train_step(batch)

            ms, ms_train = med(step3, 3), med(train3, 3)
            # train = encode + latent UNet step; sample = 35 latent evaluations + decode
            nbytes = (B * ALGO_A["ae_enc"] + W_e) + step_bytes(ALGO_A["latent_unet"], W_u, B, nfe) + (B * ALGO_A["ae_dec"] + W_d)
            r = entry(f"cfg3: 1-D latent EDM: frozen autoencoder 3x16384 <-> 16x4096 + paper-shape latent UNet, B={B}: 1 train step "
                      f"(encode + latent UNet fwd/bwd + Adam + EMA) + {args.sample_steps}-step latent Heun sample + decode",
                      B, "waveforms/s (3x16384)", ms, ms_train, nbytes)
            if cpu and B == 16:
                r["cpu_baseline"] = _cpu_extra("latent", 2, T, args, 1, 150, "latent EDM (encode + latent step; latent sample + decode), 3x16384")
            res.append(r)
            del batch
        del edm, ae, tr
    except Exception as e:
        res.append(dict(config="cfg3", error=repr(e)))
    torch.cuda.empty_cache()
    # ---- cfg4: consistency-model sampling on the paper UNet (consistency_model.py:81-106), B = 64: one network evaluation per sample
    # (sigmas = []) and the reference's default call (sigmas = [1.0]: two evaluations + one uniform-noise refinement)
    try:
        release()
        from tqdne_amd import UNetModel
        from tqdne_amd.consistency_model import LithningConsistencyModel
        torch.manual_seed(args.seed)
        net = UNetModel(**paper_1d_unet_config())
        net.load_state_dict(perturbed_state(net, 17))
        cm = LithningConsistencyModel(net).to(dev).eval()
        B, T = 64, 4096
        g = torch.Generator().manual_seed(4323)
        eps4, cond4 = torch.randn(B, 3, T, generator=g).to(dev), torch.randn(B, 5, generator=g).to(dev)
        u4 = torch.rand(B, 3, T, generator=g).to(dev)
        ms1 = med(lambda: cm.sample_from(eps4, [], [], None, cond4), 7)
        ms2 = med(lambda: cm.sample_from(eps4, [1.0], [u4], None, cond4), 7)
        r = dict(config="cfg4: consistency-model sampling (consistency_model.py:81-106) on the paper UNet, B=64, 3x4096: step = ONE "
                        "1-step sample of the batch (1 network evaluation); parts: the reference's default sigmas=[1.0] call (2 evaluations)",
                 value=B / (ms1 * 1e-3), unit="waveforms/s", ms_per_step=ms1,
                 parts=dict(one_step_sample_ms=ms1, two_step_sample_ms=ms2, two_step_waveforms_per_s=B / (ms2 * 1e-3)),
                 hbm_roofline=hbm_block(B * ALGO_A["paper"] + 4 * nparam(net), ms1))
        if cpu:
            r["cpu_baseline"] = _cpu_extra("consistency", 4, T, args, 3, 90, "paper UNet under the consistency forward, 1 network evaluation, 3x4096")
        res.append(r)
        del cm, net
    except Exception as e:
        res.append(dict(config="cfg4", error=repr(e)))
    torch.cuda.empty_cache()
    # ---- the reference's real data shape (experiments/config.py:62-67, train_1d_edm.py): 3 x 4064 waveforms -> MovingAverageEnvelope ->
    # 6 x 4064 signals, paper UNet with 6 channels in / out, B = 64; the representation and its inverse run on the GPU here
    try:
        release()
        from tqdne_amd.representation import MovingAverageEnvelope
        torch.manual_seed(args.seed)
        edm = LightningEDM(paper_1d_unet_config(in_channels=6, out_channels=6), opt, num_sampling_steps=args.sample_steps)
        edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
        edm = edm.to(dev)
        B, T = 64, 4064
        g = torch.Generator().manual_seed(4324)
        wave = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)
        cond6 = torch.randn(B, 5, generator=g).to(dev)
        rep = MovingAverageEnvelope()
        tr = DataParallelTrainer(edm, world_size=1, ema_decay=EMA_DECAY)

        def train6():
            edm.train()
            tr.train_step({"signal": rep.get_representation(wave), "cond": cond6})

        def step6():
            train6()
            edm.eval()
            return rep.invert_representation(edm.sample((B, 6, T), cond=cond6))

        def repr6():
            return rep.invert_representation(rep.get_representation(wave))

        ms, ms_train, ms_rep = med(step6, 3), med(train6, 3), med(repr6, 5)
        # (the 3 x 4096 figure scaled to 4064 positions, plus the three extra stem-input / head-output channels)
        A6 = ALGO_A["paper"] * T / 4096 + 4.0 * T * 6
        res.append(entry("reference data shape: 3x4064 waveforms through MovingAverageEnvelope (GPU) -> 6x4064, paper UNet 6 -> 6 channels, "
                         f"B=64: 1 train step (representation + fwd/bwd + Adam + EMA) + {args.sample_steps}-step Heun sample + inverse representation",
                         B, "waveforms/s (3x4064)", ms, ms_train, step_bytes(A6, 4 * nparam(edm.unet), B, nfe),
                         representation_fwd_inv_ms=ms_rep))
        del edm, tr
    except Exception as e:
        res.append(dict(config="6x4064", error=repr(e)))
    torch.cuda.empty_cache()
    # ---- the reference's default per-device batch, -b 256 (experiments/train_1d_edm.py:83-85), paper UNet 3 x 4096: the train step
    # with every activation kept and with use_checkpoint=True (block-internal activations recomputed in the backward), and the step
    for ckpt in (False, True):
        try:
            release()
            torch.manual_seed(args.seed)
            edm = LightningEDM(dict(paper_1d_unet_config(), use_checkpoint=ckpt), opt, num_sampling_steps=args.sample_steps)
            edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
            edm = edm.to(dev)
            B, T = 256, 4096
            g = torch.Generator().manual_seed(4325)
            batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev), "cond": torch.randn(B, 5, generator=g).to(dev)}
            tr = DataParallelTrainer(edm, world_size=1, ema_decay=EMA_DECAY)
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)

            def train256():
                edm.train()
                tr.train_step(batch)

            ms_train = med(train256, 3)
            peak = (torch.cuda.max_memory_allocated(dev) - base) / 2 ** 30
            W_ = 4 * nparam(edm.unet)
            r = dict(config=f"reference default batch: paper UNet, B=256, 3x4096, use_checkpoint={ckpt}: 1 train step (dropout 0.1, Adam, EMA)",
                     value=B / (ms_train * 1e-3), unit="waveforms/s (train step only)", ms_per_step=ms_train, train_wf_s=B / (ms_train * 1e-3),
                     parts=dict(train_ms=ms_train, peak_memory_gib_above_weights_and_inputs=peak),
                     hbm_roofline=hbm_block(step_bytes(ALGO_A["paper"], W_, B, nfe, sample=False), ms_train))
            if not ckpt:
                sig = edm.edm.sampling_sigmas(args.sample_steps).to(dev)
                eps0 = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev) * sig[0]

                def sample256():
                    edm.eval()
                    edm.sample_deterministically(eps0, sig, None, batch["cond"])

                ms_s = med(sample256, 2)
                r["parts"]["sample_ms"] = ms_s
                r["sample_wf_s"] = B / (ms_s * 1e-3)
                r["step_wf_s"] = B / ((ms_train + ms_s) * 1e-3)
                r["hbm_roofline_step"] = hbm_block(step_bytes(ALGO_A["paper"], W_, B, nfe), ms_train + ms_s)
                del eps0
            res.append(r)
            del edm, tr, batch
        except Exception as e:
            res.append(dict(config=f"B=256 use_checkpoint={ckpt}", error=repr(e)))
        torch.cuda.empty_cache()
    return res


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def self_launch(n):
    """``python bench.py --gpus N`` outside a launcher: start N ranks (one per GPU) as a child ``torch.distributed.run`` and exit
    with its code.  Decided before anything in this process touches the GPU (nothing is exec'ed; this process only waits)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    log("launching", n, "ranks:", " ".join(cmd))
    return subprocess.call(cmd, env=env)


_RESULT_OUT = None   # the process's ORIGINAL stdout once claim_stdout() has run (main); None: sys.stdout


def claim_stdout():
    """stdout must carry the ONE JSON line and nothing else, but libraries write there too: RCCL prints a five-line version banner
    through C stdio when its communicator is created (block-buffered on a pipe: it comes out at exit, i.e. AFTER the result line --
    seen with TQDNE_BENCH_FORCE_RCCL=1, profiles/r04_t_*), hipcc and torch warn there.  So file descriptor 1 is pointed at stderr for
    the rest of the process and the result line (or the watchdog's diagnostic line) is written to a duplicate of the original."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _RESULT_OUT


class Watchdog:
    """A hung rendezvous or first collective must be VISIBLE in the driver's record (SCALE_rNN.json) instead of a silent timeout of
    the whole run: a timer thread that, unless cancelled in time, prints ONE diagnostic JSON line on stdout (same top-level keys as
    the result line, ``value`` null, ``error`` set) and ends the process with exit code 3 -- a fresh exit, nothing is re-exec'ed."""

    def __init__(self, seconds: float, stage: str, rank: int, world: int, extra=None):
        import threading
        self.stage, self.rank, self.world, self.seconds, self.extra = stage, rank, world, seconds, dict(extra or {})
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._t.start()

    def _fire(self):
        line = dict(metric="waveforms/sec (train step + 18-step EDM sample), 3ch x 4096", value=None, unit="waveforms/s",
                    n_gpus=self.world, error=f"rank {self.rank}: stage '{self.stage}' did not complete within {self.seconds:.0f} s",
                    stage=self.stage, rank=self.rank,
                    env={k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE",
                                                         "HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES", "NCCL_DEBUG")},
                    **self.extra)
        try:
            out = _RESULT_OUT or sys.stdout
            out.write(json.dumps(line) + "\n")
            out.flush()
            log("WATCHDOG:", line["error"])
        finally:
            os._exit(3)

    def cancel(self):
        self._t.cancel()


def init_distributed(backend: str, rank: int, world: int, dev=None, init_timeout_s: float = None, first_timeout_s: float = None) -> int:
    """init_process_group + the FIRST collective (a ones tensor summed over the ranks) under a watchdog each; returns the number of
    ranks that took part, which must equal ``world``.  Timeouts: TQDNE_BENCH_INIT_TIMEOUT (default 420 s: a fresh box pages torch in
    for 1-2 minutes per rank) and TQDNE_BENCH_FIRST_COLLECTIVE_TIMEOUT (default 180 s)."""
    import datetime
    import torch.distributed as dist
    init_timeout_s = float(os.environ.get("TQDNE_BENCH_INIT_TIMEOUT", 420)) if init_timeout_s is None else init_timeout_s
    first_timeout_s = float(os.environ.get("TQDNE_BENCH_FIRST_COLLECTIVE_TIMEOUT", 180)) if first_timeout_s is None else first_timeout_s
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wd = Watchdog(init_timeout_s, "init_process_group", rank, world, dict(backend=backend))
    kw = dict(device_id=dev) if (backend == "nccl" and dev is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=init_timeout_s + first_timeout_s + 60), **kw)
    wd.cancel()
    wd = Watchdog(first_timeout_s, "first_collective", rank, world, dict(backend=backend))
    ones = torch.ones(1, device=dev if dev is not None else "cpu")
    all_reduce_(ones)
    n = int(ones.item())   # (synchronises: the collective has run)
    wd.cancel()
    if n != world:
        raise SystemExit(f"first collective summed {n} ranks, expected {world}")
    return n


def _host_staged() -> bool:
    """the dry-run backend (gloo) exchanges host tensors: device tensors are staged through the host around every collective"""
    import torch.distributed as dist
    return dist.get_backend() == "gloo"


def all_reduce_(t, op=None):
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and _host_staged():
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_gather_(t, world):
    import torch.distributed as dist
    if t.is_cuda and _host_staged():
        out = [torch.zeros_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(out, t.cpu())
        return out
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return out


def replica_checksum(params) -> "torch.Tensor":
    """(2,) int64 on the parameters' device: wrapping sum of the fp32 bit patterns and their count -- equal on two ranks iff (up to a
    2^-64 accident) the replicas hold bit-identical weights.  No host sync."""
    ps = [p.detach().reshape(-1) for p in params]
    flat = torch.cat(ps) if len(ps) > 1 else ps[0]
    bits = flat.contiguous().view(torch.int32).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64) % 1000003   # (position-weighted: a permutation changes it)
    return torch.stack([(bits * w).sum(), torch.tensor(bits.numel(), device=bits.device, dtype=torch.int64)])


def gather_checksums(cs: "torch.Tensor", world: int, collective: bool = None):
    """every rank's checksum on every rank -> list of ints (hex strings in the JSON line)"""
    import torch.distributed as dist
    if not (world > 1 if collective is None else collective):
        return [int(cs[0].item())]
    return [int(t[0].item()) for t in all_gather_(cs, world)]


def op_class(name):
    """kernel class of a plan launch, by its name (DESIGN.md section 5's table)"""
    if name.startswith("conv:"):
        if name.endswith("+skip"):
            return "resblock k5 conv + fused 1x1 skip"
        if ".qkv" in name or ".proj_out" in name:
            return "attention qkv / proj 1x1 convs"
        if name.endswith(".op") or name.endswith(".conv") or name.endswith("+polyphase"):
            return "down / up-sampling convs"
        return "resblock k5 convs"
    if name.startswith("attention"):
        return "attention core"
    if name.startswith("gn_"):
        return "GroupNorm finalise / backward"
    if name in ("embed", "stem", "head"):
        return "embedding + stem / head"
    for pfx, cls in (("wgrad:", "conv weight gradients"), ("dgrad:", "conv data gradients"), ("colsum", "bias / embedding column sums")):
        if name.startswith(pfx):
            return cls
    return "other (" + name.split(":")[0] + ")"


def class_table(trace, peak_tflops, peak_gbs):
    """per-class sums of a traced pass: time from HIP events around every launch on the launch stream"""
    torch.cuda.synchronize()
    rows = {}
    for name, flops, nbytes, e0, e1 in trace:
        r = rows.setdefault(op_class(name), dict(launches=0, ms=0.0, gflop=0.0, mbytes=0.0))
        r["launches"] += 1
        r["ms"] += e0.elapsed_time(e1)
        r["gflop"] += flops / 1e9
        r["mbytes"] += nbytes / 1e6
    total = sum(r["ms"] for r in rows.values())
    for r in rows.values():
        r["share"] = r["ms"] / total if total else None
        r["mfma_frac"] = (r["gflop"] / r["ms"] / peak_tflops) if r["ms"] and r["gflop"] else None   # GFLOP/ms = TFLOP/s
        r["hbm_frac"] = (r["mbytes"] / r["ms"] / peak_gbs) if r["ms"] and r["mbytes"] else None      # MB/ms = GB/s
        for k in ("ms", "gflop", "mbytes"):
            r[k] = round(r[k], 4)
    return dict(total_ms=total, classes=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1]: 64)")
    ap.add_argument("--length", type=int, default=4096)
    ap.add_argument("--sample-steps", type=int, default=18)
    ap.add_argument("--config", default="paper", choices=["paper", "tiny"])
    ap.add_argument("--mode", default="step", choices=["step", "train", "sample", "consistency"],
                    help="step (headline): 1 train step + one 18-step sample; train / sample: one half only (debug); "
                         "consistency: BASELINE configs[4], 1-step consistency sampling, rank-sharded, no collective")
    ap.add_argument("--no-train", action="store_true", help="= --mode sample")
    ap.add_argument("--no-sample", action="store_true", help="= --mode train")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the same-run parity gate (it rides in the CPU-baseline child)")
    ap.add_argument("--no-tables", action="store_true", help="skip the per-class traced passes after the timed region")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (HBM traffic of the dominant conv, measured in this run)")
    ap.add_argument("--no-overlap", action="store_true", help="issue the gradient all-reduce after the backward instead of under it")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg0 / cfg3 extras of the headline run")
    ap.add_argument("--no-ema", action="store_true", help="train step without the EMA of the weights (the reference's 1-D trainer "
                                                          "always keeps one: train_1d_edm.py:55; A/B switch)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", action="store_true", help="replay the whole sampler integration from one HIP graph (neutral at B=64: GPU-bound)")
    ap.add_argument("--no-auto-graph", dest="auto_graph", action="store_false",
                    help="never graph-replay the sampler (default: the sampler's own choice: eager unless TQDNE_SAMPLER_GRAPH=1)")
    args = ap.parse_args()
    if args.no_train:
        args.mode = "sample"
    if args.no_sample:
        args.mode = "train"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    result_out = claim_stdout()   # (after the self-launch: the ranks it starts inherit the real stdout)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # Dry run of the N > 1 program on a 1-GPU box: TQDNE_BENCH_BACKEND=gloo (collectives over gloo, device tensors staged through the
    # host) + TQDNE_BENCH_SHARE_DEVICE=1 (every rank on cuda:0).  Everything else is the program the driver runs on a node -- the
    # self-launch through torch.distributed.run, per-rank shards and seeds, the rank-0 broadcast, the bucketed exchange issued from inside
    # the backward sweep, the max-over-ranks timing, the checksums, exit codes 3 / 4 -- but the number is NOT a scaling number (the ranks
    # share one GPU, the exchange blocks the host) and the line says so.
    backend = os.environ.get("TQDNE_BENCH_BACKEND", "nccl").lower()
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"TQDNE_BENCH_BACKEND={backend!r}: expected nccl (RCCL) or gloo (dry run)")
    share_device = os.environ.get("TQDNE_BENCH_SHARE_DEVICE", "0") == "1"
    dev_index = 0 if share_device else local_rank
    if not share_device and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} GPU(s) visible (dry run on one GPU: TQDNE_BENCH_BACKEND=gloo "
                         "TQDNE_BENCH_SHARE_DEVICE=1)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    first_collective_ranks = 1
    # TQDNE_BENCH_FORCE_RCCL=1: at N = 1 too, create the RCCL communicator and run every collective of the N > 1 path (broadcast,
    # bucketed all-reduce from inside the backward sweep, the checksum gathers) over the one rank -- the self-test of that path
    # that a 1-GPU box allows; the JSON line says so ("rccl_forced_at_world1")
    force_rccl = world == 1 and os.environ.get("TQDNE_BENCH_FORCE_RCCL", "0") == "1"
    if force_rccl and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    multi = world > 1 or force_rccl   # (the collectives of the N > 1 path run)
    if multi and os.environ.get("TQDNE_BENCH_RESERVE_STREAMS_FIRST", "1") == "1":
        # The sampler lanes / the backward's weight-gradient stream must make their first submission BEFORE the RCCL communicator
        # exists: ROCm binds a stream to a hardware queue at its first submission, and with the communicator's streams in first the
        # backward's two streams end up sharing a queue (measured with the forced one-rank exchange: train half 22.1 -> 28.1 ms,
        # profiles/r04_u_rccl_ab.txt).  tqdne_amd.engine.reserve_side_streams is what a plan calls when it is first built.
        from tqdne_amd.engine import reserve_side_streams
        reserve_side_streams(dev)
    if multi:
        first_collective_ranks = init_distributed(backend, rank, world, dev)

    import __graft_entry__
    if rank == 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):   # (stdout carries the one JSON line only)
            __graft_entry__.build()
    if multi:
        dist.barrier()
    from tqdne_amd import LightningEDM, paper_1d_unet_config, rng, tiny_1d_unet_config
    from tqdne_amd.trainer import DataParallelTrainer
    from tqdne_amd.edm import sampler_lanes

    cfg = paper_1d_unet_config() if args.config == "paper" else tiny_1d_unet_config()
    B, T = args.batch, args.length
    do_train = args.mode in ("step", "train")
    do_sample = args.mode in ("step", "sample")
    torch.manual_seed(args.seed)  # identical initial weights on every rank (and a rank-0 broadcast in the trainer on top)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0}, num_sampling_steps=args.sample_steps)
    sd = perturbed_state(edm.unet, 17)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev)
    cm = None
    if args.mode == "consistency":
        from tqdne_amd.consistency_model import LithningConsistencyModel
        cm = LithningConsistencyModel(edm.unet).to(dev).eval()

    # every rank: its own shard of synthetic data and its own random streams (noise levels, noise, dropout masks, sampler seeds)
    rng.seed_rank(args.seed, rank)
    g = torch.Generator().manual_seed(1234 + rank)
    signal = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)
    cond = torch.randn(B, 5, generator=g).to(dev) if cfg["cond_features"] else None
    batch = {"signal": signal}
    if cond is not None:
        batch["cond"] = cond
    start_noise = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev)

    # same-run parity gate, GPU half: the HIP path on the injected inputs, on the initial weights, outside the timed region
    # (the CPU oracle child computes the other half at the end; compared in the JSON line's "parity" block)
    gpu_parity = None
    want_parity = rank == 0 and world == 1 and cm is None and not args.no_cpu_baseline and not args.no_parity
    if want_parity:
        pin = parity_inputs(cfg, args.cpu_batch, T)
        pd = {k: (None if v is None else v.to(dev)) for k, v in pin.items()}
        edm.eval()
        with torch.no_grad():
            den = edm(pd["noisy"], pd["sigma"], None, pd["cond"]).cpu()
            lss = edm.step_with_noise(pd["signal"], pd["eps"], pd["noise"], cond=pd["cond"]).cpu()
            psig = edm.edm.sampling_sigmas(args.sample_steps).to(dev)
            smp = edm.sample_deterministically(pd["start"] * psig[0], psig, None, pd["cond"]).cpu()
            # the state after k steps = the same integration over the first k + 1 sigmas (the Heun correction is skipped only on
            # step num_sampling_steps - 1, so every step of the truncated schedule is a full one, as in the oracle's trace)
            traj = {f"sample_step{k}": edm.sample_deterministically(pd["start"] * psig[0], psig[:k + 1], None, pd["cond"]).cpu()
                    for k in PARITY_TRAJECTORY if k < args.sample_steps}
        gpu_parity = dict(denoise=den, loss=lss, sample=smp, **traj)
        del pd

    ema_decay = None if args.no_ema else EMA_DECAY
    trainer = DataParallelTrainer(edm, world_size=world, overlap=not args.no_overlap, force_exchange=force_rccl,
                                  ema_decay=ema_decay) if do_train else None
    # replicas: every rank's weights right after the trainer's rank-0 broadcast (compared across ranks in the JSON line)
    cs_start = replica_checksum(edm.unet.parameters())
    sigmas = edm.edm.sampling_sigmas(args.sample_steps).to(dev)
    eps0 = start_noise * sigmas[0]
    eps32 = start_noise.float()
    use_graph = True if args.graph else (None if args.auto_graph else False)   # None: the sampler's own choice (eager unless TQDNE_SAMPLER_GRAPH=1)

    # HIP-event probe around the dominant kernel (the heaviest k=5 conv launch of the forward)
    eng = edm.unet._engine(B, T, dev)
    probe = eng.install_probe()

    def train_half():
        edm.train()
        trainer.train_step(batch)

    def sample_half():
        edm.eval()
        edm.sample_deterministically(eps0, sigmas, None, cond, use_graph=use_graph)

    def one_step():
        if cm is not None:
            cm.sample_from(eps32, [], [], None, cond)
            return
        if do_train:
            train_half()
        if do_sample:
            sample_half()

    def sync():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
            torch.cuda.synchronize(dev)

    log("model on device, plan built; warmup ...")
    for _ in range(args.warmup):
        one_step()
    sync()
    log("warmup done; timing", args.steps, "steps")
    probe.reset()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        one_step()
        marks[k + 1].record()   # (on the main stream, which every sampler lane / the exchange joins at the end of a step)
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    dt_rank = dt
    rank_ms = [1e3 * dt_rank / args.steps]
    rccl_ranks = 1
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        all_reduce_(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # self-evidence of the collective: a ones tensor summed over RCCL counts the ranks that took part, and every rank's own
        # time per step is gathered (the headline uses the MAX, as the contract asks)
        ones = torch.ones(1, device=dev)
        all_reduce_(ones)
        rccl_ranks = int(ones.item())
        allr = all_gather_(torch.tensor([1e3 * dt_rank / args.steps], device=dev, dtype=torch.float64), world)
        rank_ms = [float(t.item()) for t in allr]
    # replicas after the timed steps: data parallelism is only correct if every rank applied the same update to the same weights
    sums_start = gather_checksums(cs_start, world, multi)
    sums_end = gather_checksums(replica_checksum(edm.unet.parameters()), world, multi)
    replicas = dict(after_broadcast=len(set(sums_start)) == 1, after_timed_steps=len(set(sums_end)) == 1,
                    weights_moved=(sums_start[0] != sums_end[0]) if do_train else None,
                    checksum_after_broadcast=[f"{v & 0xFFFFFFFFFFFFFFFF:016x}" for v in sums_start],
                    checksum_after_timed_steps=[f"{v & 0xFFFFFFFFFFFFFFFF:016x}" for v in sums_end])
    ms_per_step = 1e3 * dt / args.steps
    log(f"timed region done: {ms_per_step:.1f} ms/step")
    value = world * B / (dt / args.steps)

    # separate timings of the two halves (reported, not the headline).  `<half>_ms`: 5 calls back to back between two synchronisations,
    # divided by 5 -- the half's throughput time, the way it runs inside the timed steps (no host synchronisation between launches);
    # `<half>_ms_synced`: median of 5 calls each followed by a synchronisation (rounds 1-3's definition: includes the host's launch
    # latency at the start and the drain at the end of every call, ~2 ms for the ~600 launches of a training step)
    parts = {}
    for name, fn in (("train", train_half if do_train else None), ("sample", sample_half if do_sample else None)):
        if fn is None or cm is not None:
            continue
        fn(); sync()
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            fn()
            sync()
            ts.append(1e3 * (time.perf_counter() - t1))
        parts[name + "_ms_synced"] = _median(ts)
        t1 = time.perf_counter()
        for _ in range(5):
            fn()
        sync()
        parts[name + "_ms"] = 1e3 * (time.perf_counter() - t1) / 5

    # exposed (non-overlapped) part of the gradient exchange, same run: train half with the all-reduce issued after the backward
    # minus the train half as benchmarked (issued from inside the sweep).  World 1: the exchange is a no-op, reported as 0.
    exchange = None
    if do_train and cm is None:
        exchange = dict(rccl_ranks=rccl_ranks, overlap=not args.no_overlap, bucket_bytes=4 * getattr(trainer, "bucket_elems", 0))
        if multi and not args.no_overlap:
            def med_train():
                train_half(); sync()
                ts = []
                for _ in range(5):
                    t1 = time.perf_counter()
                    train_half()
                    sync()
                    ts.append(1e3 * (time.perf_counter() - t1))
                return _median(ts)
            trainer.overlap = False
            t_after = med_train()
            trainer.overlap = True
            train_half(); sync()   # (the bucket sizes / tail words reported below are those of a step with the exchange under the backward)
            tt = torch.tensor([t_after, parts.get("train_ms_synced", 0.0)], device=dev, dtype=torch.float64)   # (both: synced medians)
            all_reduce_(tt, op=dist.ReduceOp.MAX)
            exchange.update(train_ms_exchange_after_backward=float(tt[0]), train_ms_exchange_under_backward=float(tt[1]),
                            hidden_by_overlap_ms=float(tt[0] - tt[1]), buckets_elems=list(trainer.last_bucket_sizes),
                            tail_words=int(getattr(trainer, "last_tail_words", 0)))
        else:
            exchange.update(hidden_by_overlap_ms=0.0 if not multi else None)

    tables = None
    if rank == 0 and not args.no_tables and cm is None:
        # per-class tables: one extra forward / train step with HIP events around EVERY launch (one lane, eager, outside the
        # timed region; the sampler's lanes overlap each other, so the inference pass is traced on a single B-sample lane)
        tables = {}
        eng._trace = []
        edm.eval()
        with torch.no_grad():
            edm(signal, torch.full((B,), 0.7, device=dev), None, cond)
        tables["inference_forward_1lane"] = class_table(eng._trace, MFMA_BF16_DENSE_PEAK_TFLOPS, HBM_PEAK_GBS)
        eng._trace = None
        if do_train:
            edm.train()
            edm.step_and_backward(batch)   # makes sure the backward plan exists
            eng._trace, eng._bwd._trace = [], []
            edm.step_and_backward(batch)
            tables["train_forward"] = class_table(eng._trace, MFMA_BF16_DENSE_PEAK_TFLOPS, HBM_PEAK_GBS)
            tables["train_backward"] = class_table(eng._bwd._trace, MFMA_BF16_DENSE_PEAK_TFLOPS, HBM_PEAK_GBS)
            eng._trace = eng._bwd._trace = None
    if multi:
        dist.barrier()

    if rank == 0:
        flops_fwd = conv_flops_per_sample(edm.unet, T)
        k_ms, k_flops, k_name, k_n = probe.result()
        from tqdne_amd import _lib as _tl
        scheme = os.environ.get("TQDNE_CONV_SCHEME", _tl.DEFAULT_SCHEME).lower()
        if any(e.scheme == "bf16x3" for e in edm.unet._engine_cache.values()):
            scheme = "bf16x3"   # (the range guard moved the plans)
        mult = {"f16mx6": 1.5, "f16mx8": 2.0, "bf16x3": 3.0}[scheme]
        notes = {
            "f16mx6": "fp32 product contracted as 2 fp16 MFMAs + 1 block-scaled fp6 (e2m3, per-lane E8M0 scales) MFMA per 64 channels "
                      "(TQ_WFMT_F16_MX6: the MFMA cycles of 1.5 bf16 products per algorithmic product; no TF32/xf32 on gfx950); frac = "
                      "algorithmic FLOP/s over the dense bf16 MFMA peak, so 2/3 is the ceiling of this scheme",
            "f16mx8": "fp32 product contracted as 2 fp16 MFMAs + 1 block-scaled fp8 MFMA per 64 channels (TQ_WFMT_F16_MX8: the MFMA "
                      "cycles of 2 bf16 products per algorithmic product; no TF32/xf32 on gfx950); frac = algorithmic FLOP/s over the "
                      "dense bf16 MFMA peak, so 1/2 is the ceiling of this scheme",
            "bf16x3": "fp32 operands as bf16 hi/lo, 3 MFMA products per algorithmic product (no TF32/xf32 on gfx950); frac = algorithmic "
                      "FLOP/s over the dense bf16 MFMA peak, so 1/3 is the ceiling of this scheme"}
        roofline = dict(bound="mfma", achieved=(k_flops / (k_ms * 1e-3) / 1e12) if k_ms else None,
                        peak=MFMA_BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=None, traffic=None,
                        kernel=k_name, launches_timed=k_n, avg_launch_ms=k_ms, algorithmic_flop_per_launch=k_flops,
                        executed_mfma_flop_equiv_per_launch=mult * k_flops, scheme=scheme, note=notes[scheme])
        if roofline["achieved"]:
            roofline["frac"] = roofline["achieved"] / roofline["peak"]
        # HBM traffic of the same launch from PMC counters (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected in
        # separate rocprofv3 --pmc passes by tools/pmc_dominant.sh; bench.py cannot read PMCs itself.  A PMC file recorded for
        # an older build of the conv kernel is refused (the kernel source's hash is stored in the file).
        import glob
        import hashlib
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_dominant_conv.json")))
        # (the kernel template lives in conv1d_kernel.hpp since round 4; the key keeps its round-1 name)
        src_hash = hashlib.sha256(open(os.path.join(ROOT, "tqdne_amd", "csrc", "conv1d_kernel.hpp"), "rb").read()).hexdigest()[:16]
        if pmc and args.config == "paper" and B == 64 and T == 4096:
            try:
                match = [f for f in pmc if json.load(open(f)).get("conv1d_mfma_sha16") == src_hash]   # (the newest one of THIS build)
                if match:
                    pj = json.load(open(match[-1]))
                    roofline["traffic"] = pj.get("hbm_traffic_bytes_per_launch")
                    roofline["traffic_source"] = os.path.relpath(match[-1], ROOT)
                    roofline["algorithmic_bytes_per_launch"] = pj.get("algorithmic_bytes_per_launch")
                else:
                    roofline["traffic_source"] = (os.path.relpath(pmc[-1], ROOT) + " is stale (recorded for another build of "
                                                  "the conv kernel template): traffic not reported")
            except Exception:
                pass
        if world == 1 and not args.no_pmc and args.config == "paper" and B == 64 and T == 4096 and args.mode == "step":
            # ... and measured IN THIS RUN where the profiler is there (the file above is the builder's box): two more child processes
            live = pmc_traffic_live()
            if live is not None:
                if roofline.get("traffic") is not None:
                    roofline["traffic_recorded"] = {"bytes": roofline["traffic"], "source": roofline.get("traffic_source")}
                roofline["traffic"] = live["hbm_traffic_bytes_per_launch"]
                roofline["traffic_source"] = ("this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) around "
                                              "tools/bench_one.py 256 256 256 5 1024 64 = the dominant launch alone; FETCH_SIZE x 2 (gfx950), KiB units")
                roofline["traffic_counters"] = live
                roofline["algorithmic_bytes_per_launch"] = 4 * 64 * 1024 * (512 + 256)
        nfe = 2 * args.sample_steps - 1
        n_fwd = (3 if do_train else 0) + (nfe if do_sample else 0) if cm is None else 1
        work_flop = B * flops_fwd * n_fwd
        # the north star's second fraction: fused-minimum HBM bytes of the whole step over the 8 TB/s roof.
        # Per sample and forward every conv / attention core reads its input and writes its output once in fp32 (A), weights W
        # once per call: forward = B*A + W, sample = NFE * forward, train = 3*B*A + 3*W + 7*W (Adam).  A from hooks over the
        # imported reference.
        A_W = {"paper": (ALGO_A["paper"], 62.3e6), "tiny": (ALGO_A["tiny"], 14.2e6)}.get(args.config)
        hbm_step = None
        if A_W and T == 4096:
            A_, W_ = A_W
            if cm is not None:
                algo_bytes = world * (B * A_ + W_)
            else:   # (EMA: + 3 W per train step)
                algo_bytes = world * step_bytes(A_, W_, B, nfe, train=do_train, sample=do_sample, ema=ema_decay is not None)
            gbps = algo_bytes / (dt / args.steps) / 1e9
            hbm_step = dict(bound="hbm", achieved=gbps, peak=8000.0 * world, unit="GB/s", frac=gbps / (8000.0 * world),
                            algorithmic_bytes_per_step=algo_bytes,
                            note="whole step, fused-minimum byte model of SURVEY.md 8d; the step is MFMA-bound (see roofline), "
                                 "this is the fraction the north star asks to be reported")
        if cm is not None:
            metric = "waveforms/sec (consistency 1-step sample), 3ch x 4096"
            workload = (f"{args.config} 1-D UNet under the consistency forward ({sum(p.numel() for p in edm.unet.parameters())} params), "
                        f"B={B}/GPU, 3x{T}: sample(shape, sigmas=[]) = 1 NFE; ranks sample their own seeds, no collective")
        else:
            metric = "waveforms/sec (train step + 18-step EDM sample), 3ch x 4096"
            workload = (f"{args.config} 1-D EDM UNet ({sum(p.numel() for p in edm.unet.parameters())} params), "
                        f"B={B}/GPU, 3x{T}: " + " + ".join(
                            (["1 train step (dropout 0.1, Adam, cosine LR" + (f", EMA {ema_decay}" if ema_decay is not None else "")
                              + (", gradient all-reduce " + ("under" if not args.no_overlap else "after") + " the backward" if world > 1 else "") + ")"]
                             if do_train else [])
                            + ([f"{args.sample_steps}-step Heun sample ({nfe} NFE)"] if do_sample else [])))
        out = {
            "metric": metric,
            "value": value, "unit": "waveforms/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median": _median(per_step), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (contractions on MFMA with fp32 accumulate: bf16x3, and fp16 + block-scaled-fp6 corrections on the 128/256-channel forward convs; sampler state f64)", "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}",
                       "hip_graph": bool(use_graph) or (use_graph is None and os.environ.get("TQDNE_SAMPLER_GRAPH") == "1"),
                       "sampler_lanes": 1 if (use_graph or (use_graph is None and os.environ.get("TQDNE_SAMPLER_GRAPH") == "1")) else sampler_lanes(B),
                       "mode": args.mode},
            "parts": parts,
            "parts_definition": "train_ms / sample_ms (since round 4): mean of 5 calls back to back between two synchronisations (the half's "
                                "throughput time inside the timed steps); *_ms_synced: median of 5 calls each followed by a synchronisation "
                                "(the definition train_ms / sample_ms had in rounds 1-3: ~2 ms more per training step); other_configs' parts "
                                "are synced medians",
            "collective_backend": ("rccl" if backend == "nccl" else "gloo (DRY RUN of the N > 1 program: collectives over gloo, staged "
                                   "through the host" + (", all ranks share cuda:0" if share_device else "") + " -- not a scaling number)") if multi else None,
            "shared_device": share_device,
            "rccl_forced_at_world1": force_rccl,
            "rccl_ranks": rccl_ranks, "rccl_ranks_ok": rccl_ranks == world and first_collective_ranks == world,
            "replicas_equal": replicas["after_broadcast"] and replicas["after_timed_steps"], "replicas": replicas,
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "gradient_exchange": exchange,
            "whole_step_algorithmic_tflops": work_flop / (dt / args.steps) / 1e12,
            "whole_step_mfma_frac": work_flop / (dt / args.steps) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
            "roofline": roofline,
            "hbm_roofline_whole_step": hbm_step,
        }
        # the two parts reported separately, whole job (the per-rank halves are timed one after the other on every rank)
        if "train_ms" in parts:
            out["train_wf_s"] = world * B / (parts["train_ms"] * 1e-3)
        if "sample_ms" in parts:
            out["sample_wf_s"] = world * B / (parts["sample_ms"] * 1e-3)
        if "sample_ms" in parts:
            out["whole_forward_mfma_frac"] = B * flops_fwd * nfe / (parts["sample_ms"] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS
        if "train_ms" in parts:
            out["train_mfma_frac"] = B * flops_fwd * 3 / (parts["train_ms"] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS
        if tables is not None:
            out["kernel_classes"] = tables
        if args.mode in ("train", "sample"):
            out["metric"] += " [DEBUG: partial workload, not the headline metric]"
        if multi and (backend != "nccl" or share_device):
            out["metric"] += " [DRY RUN: gloo / shared device, not a scaling number]"
        if not args.no_cpu_baseline and world == 1 and cm is None:  # (a reported baseline of the same workload: rank 0 at N = 1 only)
            log("timing the CPU oracle (bounded sample, subprocess) ...")
            import tempfile
            ppath = os.path.join(tempfile.mkdtemp(prefix="tqdne_parity_"), "cpu.npz") if gpu_parity is not None else None
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_batch, T, args.sample_steps, 99, parity_path=ppath)
            if gpu_parity is not None:
                out["parity"] = parity_block(gpu_parity, ppath, args)
        if (world == 1 and cm is None and args.mode == "step" and args.config == "paper" and B == 64 and T == 4096
                and not args.no_other_configs):
            log("other BASELINE configurations (cfg0, cfg3 B=16/64, cfg4, 6x4064, B=256) ...")
            out["other_configs"] = other_configs(dev, args)
        print(json.dumps(out), file=result_out, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rccl_ranks != world or not (replicas["after_broadcast"] and replicas["after_timed_steps"]):
        # (the line above already carries rccl_ranks_ok / replicas_equal = false; a scaling number from such a run must not pass silently)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
