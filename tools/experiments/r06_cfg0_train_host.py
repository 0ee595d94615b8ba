#!/usr/bin/env python3
"""Round 6: where does the host time of a LAUNCH-BOUND training step go?  cfg0 (tiny UNet, B = 4): the step is ~600 launches, the GPU
work ~3 ms, so every host millisecond shows.  Times the step with and without the EMA of the weights and prints the host profile."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from tqdne_amd import LightningEDM, tiny_1d_unet_config
from tqdne_amd.trainer import DataParallelTrainer

dev = torch.device("cuda:0")
B, T = 4, 4096
g = torch.Generator().manual_seed(4321)
batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)}


def make(ema):
    torch.manual_seed(0)
    edm = LightningEDM(tiny_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0}, num_sampling_steps=18)
    edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
    edm = edm.to(dev).train()
    return edm, DataParallelTrainer(edm, world_size=1, ema_decay=ema)


def med(fn, n=9):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[n // 2]


for ema in (None, 0.999, None, 0.999):
    edm, tr = make(ema)
    ms = med(lambda: tr.train_step(batch))
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step(batch)
    host = 1e3 * (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"cfg0 train step, ema={ema}: {ms:.2f} ms synced median; host time per step (20 back to back, no sync) {host:.2f} ms")
edm, tr = make(0.999)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.train_step(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
