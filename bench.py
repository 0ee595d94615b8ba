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


def _cpu_baseline_worker(cfg_name, B, T, nsample_steps, seed, nthreads):
    """Runs in a child process: time the CPU oracle (oracle/ = our PyTorch-CPU restatement, a "port") on a bounded
    sample of the same workload: one full train step (fwd + bwd + Adam) and one 18-step sample at batch B."""
    import torch
    from oracle import edm as OE
    from tqdne_amd import UNetModel, paper_1d_unet_config, tiny_1d_unet_config

    torch.set_num_threads(nthreads)
    cfg = paper_1d_unet_config() if cfg_name == "paper" else tiny_1d_unet_config()
    torch.manual_seed(0)
    sd = perturbed_state(UNetModel(**cfg), 17)
    g = torch.Generator().manual_seed(seed)
    sig = 0.5 * torch.randn(B, 3, T, generator=g)
    cond = torch.randn(B, 5, generator=g) if cfg.get("cond_features") else None
    params = {("unet." + k): v.clone().requires_grad_(v.is_floating_point() and k != "time_embed.W") for k, v in sd.items()}
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=1e-4)
    p = OE.EDMParams()
    net = OE.make_net(params, cfg)
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = OE.loss_step(p, net, sig, torch.randn(B, generator=g), torch.randn(B, 3, T, generator=g), cond=cond)
    loss.backward()
    opt.step()
    t_train = time.perf_counter() - t0
    print(json.dumps(dict(stage="train", t_train=t_train)), flush=True)
    t0 = time.perf_counter()
    with torch.no_grad():
        OE.sample_deterministic(p, net, torch.randn(B, 3, T, generator=g, dtype=torch.float64), nsample_steps, cond=cond)
    t_sample = time.perf_counter() - t0
    print(json.dumps(dict(stage="done", t_train=t_train, t_sample=t_sample)), flush=True)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg_name, B, T, nsample_steps, seed, timeout_s=240):
    """Launch the worker with a hard timeout (a slow or oversubscribed host must not stall the bench)."""
    import subprocess
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    nthreads = max(1, min(ncores, 64))
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"bench._cpu_baseline_worker({cfg_name!r}, {B}, {T}, {nsample_steps}, {seed}, {nthreads})")
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(nthreads))
    t_train = t_sample = None
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=timeout_s, env=env, cwd=ROOT)
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    for l in lines:
        t_train = l.get("t_train", t_train)
        t_sample = l.get("t_sample", t_sample)
    res = dict(value=None, unit="waveforms/s", cores=nthreads, kind="port", train_s=t_train, sample_s=t_sample,
               sample=f"{cfg_name} UNet, B={B}, 3x{T}: 1 train step + 1 x {nsample_steps}-step sample, torch {torch.__version__} CPU, "
                      f"{nthreads} threads ({ncores} cores visible, {_cpu_model()})")
    if t_train is not None and t_sample is not None:
        res["value"] = B / (t_train + t_sample)
    else:
        res["sample"] += f" -- did not finish within {timeout_s} s"
    return res


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1]: 64)")
    ap.add_argument("--length", type=int, default=4096)
    ap.add_argument("--sample-steps", type=int, default=18)
    ap.add_argument("--config", default="paper", choices=["paper", "tiny"])
    ap.add_argument("--no-train", action="store_true", help="debug: time the sampler only (NOT the headline metric)")
    ap.add_argument("--no-sample", action="store_true", help="debug: time the train step only (NOT the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--graph", action="store_true", help="replay the UNet forward of the sampler from a HIP graph (neutral at B=64: GPU-bound)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from tqdne_amd import LightningEDM, paper_1d_unet_config, tiny_1d_unet_config
    from tqdne_amd.trainer import DataParallelTrainer
    from tqdne_amd.edm import sampler_lanes

    cfg = paper_1d_unet_config() if args.config == "paper" else tiny_1d_unet_config()
    B, T = args.batch, args.length
    torch.manual_seed(0)
    edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0}, num_sampling_steps=args.sample_steps)
    sd = perturbed_state(edm.unet, 17)
    edm.unet.load_state_dict(sd)
    edm = edm.to(dev)

    g = torch.Generator().manual_seed(1234 + rank)
    signal = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)
    cond = torch.randn(B, 5, generator=g).to(dev) if cfg["cond_features"] else None
    batch = {"signal": signal}
    if cond is not None:
        batch["cond"] = cond
    start_noise = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev)

    trainer = DataParallelTrainer(edm, world_size=world) if not args.no_train else None
    sigmas = edm.edm.sampling_sigmas(args.sample_steps).to(dev)
    eps0 = start_noise * sigmas[0]
    use_graph = args.graph

    # HIP-event probe around the dominant kernel (the heaviest k=5 conv launch of the forward)
    eng = edm.unet._engine(B, T, dev)
    probe = eng.install_probe()

    def one_step():
        if trainer is not None:
            edm.train()
            trainer.train_step(batch)
        if not args.no_sample:
            edm.eval()
            edm.sample_deterministically(eps0, sigmas, None, cond, use_graph=use_graph)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    log("model on device, plan built; warmup ...")
    for _ in range(args.warmup):
        one_step()
    sync()
    log("warmup done; timing", args.steps, "steps")
    probe.reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    log(f"timed region done: {ms_per_step:.1f} ms/step")
    value = world * B / (dt / args.steps)

    # separate timings of the two halves (reported, not the headline)
    parts = {}
    if rank == 0 or world > 1:
        for name, fn in (("train", (lambda: (edm.train(), trainer.train_step(batch))) if trainer else None),
                         ("sample", (lambda: (edm.eval(), edm.sample_deterministically(eps0, sigmas, None, cond, use_graph=use_graph)))
                          if not args.no_sample else None)):
            if fn is None:
                continue
            fn(); sync()
            t1 = time.perf_counter()
            for _ in range(2):
                fn()
            sync()
            parts[name + "_ms"] = 1e3 * (time.perf_counter() - t1) / 2

    if rank == 0:
        flops_fwd = conv_flops_per_sample(edm.unet, T)
        k_ms, k_flops, k_name, k_n = probe.result()
        mx8 = os.environ.get("TQDNE_CONV_SCHEME", "f16mx8").lower() == "f16mx8"
        roofline = dict(bound="mfma", achieved=(k_flops / (k_ms * 1e-3) / 1e12) if k_ms else None,
                        peak=MFMA_BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=None, traffic=None,
                        kernel=k_name, launches_timed=k_n, avg_launch_ms=k_ms, algorithmic_flop_per_launch=k_flops,
                        executed_mfma_flop_equiv_per_launch=(2 if mx8 else 3) * k_flops,
                        note=("fp32 product contracted as 2 fp16 MFMAs + 1 block-scaled fp8 MFMA per 64 channels (TQ_WFMT_F16_MX8: "
                              "the MFMA cycles of 2 bf16 products per algorithmic product; no TF32/xf32 on gfx950); frac = algorithmic "
                              "FLOP/s over the dense bf16 MFMA peak, so 1/2 is the ceiling of this scheme") if mx8 else
                             ("fp32 operands as bf16 hi/lo, 3 MFMA products per algorithmic product (no TF32/xf32 on gfx950); "
                              "frac = algorithmic FLOP/s over the dense bf16 MFMA peak, so 1/3 is the ceiling of this scheme"))
        if roofline["achieved"]:
            roofline["frac"] = roofline["achieved"] / roofline["peak"]
        # HBM traffic of the same launch from PMC counters (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected in
        # separate rocprofv3 --pmc passes by tools/pmc_dominant.sh; bench.py cannot read PMCs itself
        import glob
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_dominant_conv.json")))
        if pmc and args.config == "paper" and B == 64 and T == 4096:
            try:
                pj = json.load(open(pmc[-1]))
                roofline["traffic"] = pj.get("hbm_traffic_bytes_per_launch")
                roofline["traffic_source"] = os.path.relpath(pmc[-1], ROOT)
                roofline["algorithmic_bytes_per_launch"] = pj.get("algorithmic_bytes_per_launch")
            except Exception:
                pass
        nfe = 2 * args.sample_steps - 1
        work_flop = B * flops_fwd * ((3 if trainer else 0) + (nfe if not args.no_sample else 0))
        # the north star's second fraction: fused-minimum HBM bytes of the whole step over the 8 TB/s roof.
        # Per sample and forward every conv / attention core reads its input and writes its output once in fp32 (A), weights W
        # once per call: forward = B*A + W, sample = NFE * forward, train = 3*B*A + 3*W + 7*W (Adam).  A from hooks over the
        # imported reference.
        A_W = {"paper": (174.7e6, 62.3e6), "tiny": (74.3e6, 14.2e6)}.get(args.config)
        hbm_step = None
        if A_W and T == 4096:
            A_, W_ = A_W
            algo_bytes = world * (((3 * B * A_ + 10 * W_) if trainer else 0) + ((nfe * (B * A_ + W_)) if not args.no_sample else 0))
            gbps = algo_bytes / (dt / args.steps) / 1e9
            hbm_step = dict(bound="hbm", achieved=gbps, peak=8000.0 * world, unit="GB/s", frac=gbps / (8000.0 * world),
                            algorithmic_bytes_per_step=algo_bytes,
                            note="whole step, fused-minimum byte model of SURVEY.md 8d; the step is MFMA-bound (see roofline), "
                                 "this is the fraction the north star asks to be reported")
        out = {
            "metric": "waveforms/sec (train step + 18-step EDM sample), 3ch x 4096",
            "value": value, "unit": "waveforms/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (contractions on MFMA with fp32 accumulate: bf16x3, and fp16 + block-scaled-fp8 corrections on the 128/256-channel forward convs; sampler state f64)", "data": "synthetic",
            "config": {"workload": f"{args.config} 1-D EDM UNet ({sum(p.numel() for p in edm.unet.parameters())} params), "
                                   f"B={B}/GPU, 3x{T}: 1 train step (dropout 0.1, Adam, cosine LR) + {args.sample_steps}-step "
                                   f"Heun sample ({nfe} NFE)", "global_batch": world * B, "parallelism": f"dp{world}",
                       "hip_graph": use_graph, "sampler_lanes": 1 if use_graph else sampler_lanes(B)},
            "parts": parts,
            "whole_step_algorithmic_tflops": work_flop / (dt / args.steps) / 1e12,
            "roofline": roofline,
            "hbm_roofline_whole_step": hbm_step,
        }
        if args.no_train or args.no_sample:
            out["metric"] += " [DEBUG: partial workload, not the headline metric]"
        if not args.no_cpu_baseline and world == 1:  # (a reported baseline of the same workload: rank 0 at N = 1 only)
            log("timing the CPU oracle (bounded sample, subprocess) ...")
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_batch, T, args.sample_steps, 99)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
